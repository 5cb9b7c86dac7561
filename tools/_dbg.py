import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import numpy as np, torch
import helpers as H, cases as CS
from lantern_amd import ops
from test_gpu_parity import dev, hip_cfg, table_dev
SPECS=H.ep_specs()
i=0
spec, case = SPECS[i], H.ep_case(i)
tb, g = H.static_inputs(spec, case)
m = CS.MODELS[spec["model"]]; lo, W = m["img_lo"], m["img_hi"]-m["img_lo"]
N = len(tb["tree_indices"]); nl = g["node_logits"]
for nrej in (1,2,3):
    uni = case["uniforms"].copy(); uni[2:2+nrej] = 0.9999; uni[2+nrej:] = 0.0
    aux = ops.StaticAux(cart_prob=dev(case["cart_prob"])[None], orig_prob=dev(g["orig_prob"])[None], op_off=dev(g["op_off"]), p_idx=dev(tb["p_indices"]), b_off=dev(tb["b_off"]), b_idx=dev(tb["b_idx"]), tree_cand=dev(case["tree_cand"])[None])
    ri = dev(H.row_index_from_retrieve(tb["retrieve"], N))
    out = ops.evaluate_posterior_window(hip_cfg(spec), m["V"], dev(nl[:, lo:lo+W])[None], lo, ri, dev(case["cand"])[None], dev(uni)[None], table=table_dev(m["K"]), aux=aux, want_dense=True)
    d = ops.evaluate_posterior(hip_cfg(spec), dev(nl)[None], ri, dev(case["cand"])[None], dev(uni)[None], table=table_dev(m["K"]), aux=aux)
    print(nrej, 'win', int(out['best'][0]), int(out['accept_len'][0]), out['counters'][0].tolist(), 'dense', int(d[0][0]), int(d[1][0]), d[3][0].tolist(), 'maxdiff', float((out['sample_p'][0]-d[2][0]).abs().max()))
print(tb['b_off'][:40], tb['b_idx'][:20], tb['p_indices'][:6])
