#!/usr/bin/env python3
"""Reference runs at BASELINE's REAL sizes: the reference's own
``evaluate_posterior`` / ``evaluate_posterior_v1`` / logits processors called on CPU tensors with

    Lumina   V = 65536, K = 8192, ids + 4, k = 1000, delta = 0.1 and lambda = 5, static mc_sim_7b_63 + dynamic N = 59
    LlamaGen V = K = 16384, top_k 2000 processors inside, static naive_extend_57 + dynamic
    Anole    V = 65536, K = 8192, static naive_extend_57 with run.sh's (k, lambda) pairs + dynamic

and a neighbour table built by the ``generate_codebook.py:53-65`` recipe on a seeded codebook
(``cases.build_table_full``; held here against ``torch.cdist`` + ``topk`` in float64).

Only the build container runs this (``/root/reference`` is imported, nothing of it is copied):

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_fullsize.py

The fixture ``evaluate_posterior_full.npz`` stores the small tensors the reference consumed, the seeds of the big ones
(+ a float64 checksum of each) and the reference's outputs: ``best, accept_len, n_draws`` and ``sample_p`` in sparse form
(ids + values of its support -- at most 2000 entries behind a top-k filter --, or dense when the support is wider).
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases as CS  # noqa: E402
import make_golden as MG  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def table_check(K, C, table):
    """The reference recipe verbatim (generate_codebook.py:53-65) in float64 on the same codebook."""
    cb = torch.from_numpy(CS.full_codebook(K, C)).double()
    d = torch.cdist(cb, cb)
    d.fill_diagonal_(float("inf"))
    _, idx = torch.topk(d, K - 1, dim=-1, largest=False)
    same = float((idx.numpy().astype(np.uint16) == table).mean())
    # float32 cdist (what a GPU run of the reference script computes): how many positions agree
    d32 = torch.cdist(cb.float(), cb.float())
    d32.fill_diagonal_(float("inf"))
    _, idx32 = torch.topk(d32, K - 1, dim=-1, largest=False)
    same32 = float((idx32.numpy().astype(np.uint16) == table).mean())
    return same, same32


def sparse(p: np.ndarray, prefix: str, out: dict):
    ids = np.flatnonzero(p != 0)
    if len(ids) <= 4096:
        out[prefix + "_ids"] = ids.astype(np.int32)
        out[prefix + "_vals"] = p[ids].astype(np.float32)
        out[prefix + "_len"] = np.int64(p.size)
    else:
        out[prefix] = p.astype(np.float32)
    out[prefix + "_sum"] = np.float64(p.astype(np.float64).sum())


def full_specs():
    specs, sid = [], 0

    def add(**kw):
        nonlocal sid
        sid += 1
        specs.append(dict(size="full", seed=9000 + sid, gen_top_k=2000, **kw))

    for (lantern, k, delta) in [(True, 1000, 0.1), (True, 1000, 5.0)]:
        for sigma in (0.5, 1.5):
            add(kind="static", model="lumina", tree="mc_sim_7b_63", lantern=lantern, k=k, delta=delta, sigma=sigma)
            add(kind="dynamic", model="lumina", lantern=lantern, k=k, delta=delta, depth=5)
    add(kind="static", model="lumina", tree="mc_sim_7b_63", lantern=True, k=10, delta=5.0, sigma=1.0)      # run.sh:76-91
    add(kind="static", model="lumina", tree="mc_sim_7b_63", lantern=False, k=1000, delta=0.1, sigma=1.0)
    add(kind="static", model="lumina", tree="naive_extend_57", lantern=True, k=1000, delta=0.1, sigma=1.0)
    for (k, delta) in [(1000, 0.1), (1000, 5.0)]:
        add(kind="static", model="llamagen", tree="naive_extend_57", lantern=True, k=k, delta=delta, sigma=1.0, top_k=2000)
        add(kind="dynamic", model="llamagen", lantern=True, k=k, delta=delta, depth=4, top_k=2000)
    add(kind="dynamic", model="llamagen", lantern=False, k=1000, delta=0.1, depth=4, top_k=2000)          # C2: standard verify
    for (k, delta) in [(5, 10.0), (10, 5.0), (5, 20.0), (1000, 0.1)]:                                     # C4 / run.sh
        add(kind="static", model="anole", tree="naive_extend_57", lantern=True, k=k, delta=delta, sigma=1.0, top_k=2000)
    add(kind="dynamic", model="anole", lantern=True, k=1000, delta=5.0, depth=4, top_k=2000)
    add(kind="greedy", model="llamagen", lantern=True, k=100, delta=0.3)
    add(kind="greedy", model="anole", lantern=True, k=1000, delta=0.1)
    return specs


def golden_o7_full(R, data):
    """O7 at V = 65536: MultiModalLogitsProcessor + InterleavedTopKLogitsWarper(2000) on a 48 x 48 grid
    (ea_model_lumina_mgpt.py:45-112, called as tree_decoding does at :597-605), f32 and bf16 inputs."""
    m = CS.FULL["lumina"]
    V = m["V"]
    rs = np.random.RandomState(4242)
    w = h = 48
    N = 12
    cond = (4 * rs.standard_normal((N, V))).astype(np.float32)
    unc = (4 * rs.standard_normal((N, V))).astype(np.float32)
    proc = R.lum.MultiModalLogitsProcessor.__new__(R.lum.MultiModalLogitsProcessor)
    proc.image_next_line_token_id = m["syntax"][2]
    proc.image_end_token_id = m["syntax"][0]
    supp = torch.ones(V, dtype=torch.bool)
    supp[m["img_lo"]:m["img_hi"]] = False
    proc.suppress_token_mask = supp
    warp = R.lum.InterleavedTopKLogitsWarper(image_top_k=2000)
    img_start = 37
    # grid rows, two newline rows (n % 49 == 48), the final row (eos), the first and the last grid position
    t = [0, 1, 47, 48, 49, 97, 500, 1175, 2350, 2351, 2352, 1000]
    pos = np.array([img_start + 3 + x for x in t], np.int64)
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        c, u = torch.from_numpy(cond).to(dt), torch.from_numpy(unc).to(dt)
        cfg = u + 3.0 * (c - u)
        x = proc(cfg, h_latent_dim=h, w_latent_dim=w, image_start_token_id_index=img_start, position_ids=torch.from_numpy(pos))
        x = warp(x).float().numpy()
        fin = np.isfinite(x)
        data[f"o7.{tag}_count"] = fin.sum(1).astype(np.int32)
        data[f"o7.{tag}_sum"] = np.where(fin, x, 0).astype(np.float64).sum(1)
        data[f"o7.{tag}_support_sha"] = np.asarray([sha(np.flatnonzero(r).astype(np.int32)) for r in fin])
        data[f"o7.{tag}_chk"] = np.float64(CS.checksum(x))
    data["o7.pos"] = pos
    data["o7.img_start"] = np.int64(img_start)
    data["o7.seed"] = np.int64(4242)
    data["o7.chk_cond"] = np.float64(CS.checksum(cond))
    print("o7 full ok: finite per row", data["o7.f32_count"].tolist())


def golden_kv_full(R, data):
    """O9 + O10 at the 7B slab geometry: the reference's update_inference_inputs (ea_model_lumina_mgpt.py:731-799, parallel CFG: one [64, 2, 32, S, 128] slab) with
    the default tree's retrieve rows; the moved slab is stored as a SHA-256."""
    cfg = types.SimpleNamespace(num_hidden_layers=32, num_key_value_heads=32, max_position_embeddings=96, hidden_size=4096, num_attention_heads=32)
    lin = types.SimpleNamespace(weight=torch.zeros(1))
    layer = types.SimpleNamespace(self_attn=types.SimpleNamespace(q_proj=lin))
    fake = types.SimpleNamespace(config=cfg, dtype=torch.float32, model=types.SimpleNamespace(layers=[layer] * 32))
    pkv, data_list, cur = R.kvc.initialize_past_key_values(fake, batch_size=2)
    slab = data_list[0]
    assert tuple(slab.shape) == (64, 2, 32, 96, 128)
    slab.copy_(torch.from_numpy(CS.kv_full_inputs()))
    tb = R.lum.generate_tree_buffers(R.ch.mc_sim_7b_63, device="cpu")
    retrieve = tb["retrieve_indices"]
    best, alen, prev = 7, 3, 41
    rs = np.random.RandomState(78)
    cand = torch.from_numpy(rs.randint(4, 8196, size=tuple(retrieve.shape)).astype(np.int64))
    hid = torch.from_numpy(rs.standard_normal((1, 26, 64)).astype(np.float32))
    uhid = torch.from_numpy(rs.standard_normal((1, 26, 64)).astype(np.float32))
    sample_p = torch.zeros(65536)
    sample_p[4321] = 1.0
    captured = {}
    ns = types.SimpleNamespace(cfg_mode="parallel", ea_layer=types.SimpleNamespace(topK_generate=lambda **kw: captured.update(kw)),
                               base_model=types.SimpleNamespace(lm_head=None), drafter_logits_processors=None, eagle_version=1)
    input_ids = torch.zeros(2, prev, dtype=torch.long)
    new_ids, _, new_token, token = R.lum.EaLumina_mGPT.update_inference_inputs(
        ns, input_ids, None, cand, torch.tensor(best), alen, retrieve, True, 0, data_list, cur, hid, uhid, sample_p)
    data.update({"kv.best": np.int32(best), "kv.accept_len": np.int32(alen), "kv.prev": np.int64(prev), "kv.retrieve": retrieve.numpy(), "kv.cand": cand.numpy(),
                 "kv.after_sha256": np.asarray(sha(slab.numpy())), "kv.before_sha256": np.asarray(sha(CS.kv_full_inputs())),
                 "kv.current_length": cur.numpy(), "kv.new_ids_tail": new_ids.numpy()[0, prev:], "kv.token": token.numpy(),
                 "kv.accept_hidden": captured["hidden_states"].numpy(), "kv.hidden": hid.numpy()})
    print("kv full ok:", sha(slab.numpy())[:16], "moved rows", retrieve[best, :alen + 1].tolist(), "->", list(range(prev, prev + alen + 1)))


def main():
    out = HERE
    torch.set_num_threads(8)
    R = MG.import_reference()
    trees = {nm: getattr(R.ch, nm) for nm in ("mc_sim_7b_63", "naive_extend_57")}
    tables, data = {}, {}

    def table_cache(spec):
        m = CS.model_dims(spec)
        key = (m["K"], m["C"])
        if key not in tables:
            tables[key] = CS.build_table_full(m["K"], m["C"])
            same, same32 = table_check(m["K"], m["C"], tables[key])
            data[f"table.{m['K']}x{m['C']}.sha256"] = np.asarray(sha(tables[key]))
            data[f"table.{m['K']}x{m['C']}.equal_to_torch_f64_recipe"] = np.float64(same)
            data[f"table.{m['K']}x{m['C']}.equal_to_torch_f32_recipe"] = np.float64(same32)
            print("table", key, "sha", sha(tables[key])[:16], "== torch f64 recipe at", same, "of the positions; f32:", same32)
            assert same > 0.999999
        return tables[key]

    specs = full_specs()
    for i, spec in enumerate(specs):
        if spec["kind"] == "static":
            res = MG.run_static_case(R, spec, trees[spec["tree"]], table_cache)
        elif spec["kind"] == "dynamic":
            res = MG.run_dynamic_case(R, spec, table_cache)
        else:
            res = MG.run_greedy_case(R, spec, table_cache)
        for key, val in res.items():
            if key == "sample_p":
                sparse(val, f"c{i}.sample_p", data)
            elif key == "out_row":            # greedy: the raw logits row of the regenerated input
                data[f"c{i}.out_row_chk"] = np.float64(CS.checksum(val))
                data[f"c{i}.out_row_sha"] = np.asarray(sha(val.astype(np.float32)))
            else:
                data[f"c{i}.{key}"] = val
        print(i, json.dumps(spec), "-> best", int(res["best"]), "alen", int(res["accept_len"]), "draws", int(res.get("n_draws", -1)),
              flush=True)
    data["specs"] = np.asarray(json.dumps(specs))
    golden_o7_full(R, data)
    golden_kv_full(R, data)
    np.savez_compressed(os.path.join(out, "evaluate_posterior_full.npz"), **data)
    print("evaluate_posterior_full.npz:", len(specs), "cases,", os.path.getsize(os.path.join(out, "evaluate_posterior_full.npz")), "bytes")


if __name__ == "__main__":
    main()
