import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo")))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden"))
import numpy as np, torch
import cases as CS, helpers as H
from lantern_amd import ops
from test_gpu_parity import hip_cfg, table_dev
from test_gpu_window import window_of
from test_gpu_nodes import _prob_rows, dev
SPECS = H.ep_specs()
for i in range(0, 12):
    spec, case = SPECS[i], H.ep_case(i)
    if spec["kind"] != "static": continue
    tb, g = H.static_inputs(spec, case)
    m = CS.MODELS[spec["model"]]
    lo, W = window_of(spec["model"])
    N = len(tb["tree_indices"])
    pr = _prob_rows(spec, g["node_logits"], lo, W)
    cfg = hip_cfg(spec); cfg.temperature, cfg.top_k, cfg.top_p = 1.0, 0, 1.0
    aux = ops.StaticAux(cart_prob=dev(case["cart_prob"])[None], orig_prob=dev(g["orig_prob"])[None], op_off=dev(g["op_off"]),
                        p_idx=dev(tb["p_indices"]), b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"] if len(tb["b_idx"]) else np.zeros(1, np.int32)), tree_cand=dev(case["tree_cand"])[None])
    nt = ops.tree_node_tables(tb["retrieve"], N, tb["p_indices"], tb["b_off"], g["op_off"], device="cuda", b_idx=tb["b_idx"])
    args = (cfg, m["V"], pr[None], lo, dev(H.row_index_from_retrieve(tb["retrieve"], N)), dev(case["cand"])[None], dev(case["uniforms"])[None])
    kw = dict(table=table_dev(m["K"]), aux=aux, u_bonus=dev(np.array([0.3])), want_dense=True, rows_probs=True)
    chain = ops.evaluate_posterior_window(*args, **kw)
    for lw in (0, 1):
        node = ops.evaluate_posterior_window(*args, nodes=nt, leaf_workgroups=lw, **kw)
        d = (node["sample_p"] - chain["sample_p"]).abs()
        print(i, spec["model"], spec.get("special"), "lw", lw, "cnt", node["counters"][0].tolist(), chain["counters"][0].tolist(), "ndiff", int((d > 0).sum()), "max", float(d.max()),
              "best", int(node["best"][0]), int(chain["best"][0]), "sum", float(node["sample_p"].sum()), float(chain["sample_p"].sum()))
