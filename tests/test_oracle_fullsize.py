"""CPU: the oracle against REFERENCE runs at BASELINE's real sizes (tests/golden/make_golden_fullsize.py ->
evaluate_posterior_full.npz): V = 65536 / K = 8192 / k = 1000 (Lumina, Anole), V = K = 16384 (LlamaGen), the neighbour
table by the generate_codebook.py recipe at its real shape.  Same bar as the reduced-size vectors of test_oracle_golden.py:
integers bit-exact, probabilities within 1e-6."""
import hashlib

import numpy as np
import pytest
import torch

import cases as CS
import helpers as H
import oracle

SPECS = H.full_specs()


def _ids(kind):
    return [i for i, s in enumerate(SPECS) if s["kind"] == kind]


def test_the_fixture_covers_the_sizes_the_small_vectors_do_not():
    d = H.load("evaluate_posterior_full.npz")
    assert float(d["table.8192x256.equal_to_torch_f64_recipe"]) == 1.0            # generate_codebook.py:53-65 in float64
    assert float(d["table.16384x8.equal_to_torch_f64_recipe"]) > 0.999999         # (exact distance ties may swap)
    lum = [s for s in SPECS if s["model"] == "lumina" and s["k"] == 1000 and s["lantern"]]
    assert {(s["kind"], s["delta"]) for s in lum} >= {("static", 0.1), ("static", 5.0), ("dynamic", 0.1), ("dynamic", 5.0)}
    assert len(lum) >= 6
    assert any(s["model"] == "llamagen" for s in SPECS) and any(s["model"] == "anole" and s.get("tree") == "naive_extend_57" for s in SPECS)
    # ids beyond 4095 really occur in what the reference consumed
    assert max(int(H.full_case(i)["cand"].max()) for i in _ids("static")) > 4096


def _check(best, alen, sp, cnt, case):
    assert best == int(case["best"])
    assert alen == int(case["accept_len"])
    assert cnt[3] == int(case["n_draws"])
    np.testing.assert_allclose(sp, case["sample_p"], rtol=0, atol=1e-6)
    assert abs(float(sp.astype(np.float64).sum()) - float(case["sample_p_sum"])) < 1e-4
    assert np.array_equal(np.flatnonzero(sp > 1e-6), np.flatnonzero(case["sample_p"] > 1e-6))


@pytest.mark.parametrize("i", _ids("static"))
def test_evaluate_posterior_static_full_size(i):
    spec, case = SPECS[i], H.full_case(i)
    tb, g = H.static_inputs(spec, case)
    cand, cprob, tcand = oracle.gather_candidates(case["ss_token"], case["ss_prob"], int(case["sample_token"]),
                                                  tb["tree_indices"], tb["retrieve"])
    assert np.array_equal(cand, case["cand"]) and np.array_equal(tcand, case["tree_cand"]) and np.array_equal(cprob, case["cart_prob"])
    N = len(tb["tree_indices"])
    best, alen, sp, cnt = oracle.evaluate_posterior(
        H.ep_config(spec), g["node_logits"], H.row_index_from_retrieve(tb["retrieve"], N), case["cand"],
        case["uniforms"], table=H.table_for(spec), aux=H.static_aux(tb, g, case))
    _check(best, alen, sp, cnt, case)


@pytest.mark.parametrize("i", _ids("dynamic"))
def test_evaluate_posterior_dynamic_full_size(i):
    spec, case = SPECS[i], H.full_case(i)
    nl, uniforms = H.dynamic_node_logits(spec, case)
    N = len(case["draft_tokens"])
    best, alen, sp, cnt = oracle.evaluate_posterior(
        H.ep_config(spec), nl, H.row_index_from_retrieve(case["retrieve"], N), case["cand"], uniforms, table=H.table_for(spec))
    _check(best, alen, sp, cnt, case)


@pytest.mark.parametrize("i", _ids("greedy"))
def test_evaluate_posterior_greedy_full_size(i):
    spec, case = SPECS[i], H.full_case(i)
    nl, _ = H.dynamic_node_logits(spec, case, greedy=True)
    N = len(case["draft_tokens"])
    m = CS.model_dims(spec)
    best, alen, row = oracle.evaluate_posterior_greedy(
        nl, H.row_index_from_retrieve(case["retrieve"], N), case["cand"], lantern=spec["lantern"], k=spec["k"],
        delta=spec["delta"], tok_offset=m["off"], table=H.table_for(spec))
    assert (best, alen) == (int(case["best"]), int(case["accept_len"]))
    assert hashlib.sha256(np.ascontiguousarray(row, np.float32).tobytes()).hexdigest() == str(case["out_row_sha"])


@pytest.mark.parametrize("i", _ids("dynamic")[::2])
def test_dynamic_tree_at_full_vocabulary(i):
    """O3 + O4 on [10, 65536] / [10, 16384] drafter rows: the tree the reference's topK_genrate built."""
    spec, case = SPECS[i], H.full_case(i)
    depth, k = int(case["depth"]), CS.TOPK
    script = H.dynamic_script(spec["seed"], spec["model"], depth, m=CS.model_dims(spec))
    assert abs(sum(CS.checksum(s) for s in script) - float(case["chk_script"])) < 1e-6
    ti, cu, ci, scores = oracle.expand_dynamic(H.hf_process_rows(script[0][None], H.DYN_TOP_K), None, k)
    scores_list, tokens_list, parents_list = [cu.reshape(-1)], [ti.reshape(-1)], [np.zeros(1, np.int64)]
    topk_cs_index = np.arange(k)
    for d in range(depth):
        parents_list.append(topk_cs_index + 1 + k * k * max(0, d - 1) + (k if d > 0 else 0))
        ti, cu, ci, scores = oracle.expand_dynamic(H.hf_process_rows(script[d + 1], H.DYN_TOP_K), scores, k)
        topk_cs_index = ci
        scores_list.append(cu.reshape(-1))
        tokens_list.append(ti.reshape(-1))
    draft, retrieve, mask, pos = oracle.tree_dynamic_finalize(
        np.concatenate(scores_list), np.concatenate(tokens_list), np.concatenate(parents_list), k,
        int(case["total_tokens"]), int(case["sample_token"]), sort_rows=True)
    assert np.array_equal(draft, case["draft_tokens"]) and np.array_equal(retrieve, case["retrieve"])
    assert np.array_equal(mask, case["mask"]) and np.array_equal(pos, case["pos"])


def o7_full_inputs():
    d = H.load("evaluate_posterior_full.npz")
    rs = np.random.RandomState(int(d["o7.seed"]))
    V = CS.FULL["lumina"]["V"]
    cond = (4 * rs.standard_normal((12, V))).astype(np.float32)
    unc = (4 * rs.standard_normal((12, V))).astype(np.float32)
    assert abs(CS.checksum(cond) - float(d["o7.chk_cond"])) < 1e-6
    return d, cond, unc


def o7_full_check(out, d, tag):
    fin = np.isfinite(out)
    assert np.array_equal(fin.sum(1), d[f"o7.{tag}_count"])
    got = [hashlib.sha256(np.flatnonzero(r).astype(np.int32).tobytes()).hexdigest() for r in fin]
    assert got == [str(x) for x in d[f"o7.{tag}_support_sha"]]
    np.testing.assert_allclose(np.where(fin, out, 0).astype(np.float64).sum(1), d[f"o7.{tag}_sum"], rtol=0, atol=1e-9)
    assert abs(CS.checksum(out) - float(d[f"o7.{tag}_chk"])) < 1e-9


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_cfg_mask_topk_full_size(tag):
    d, cond, unc = o7_full_inputs()
    m = CS.FULL["lumina"]
    bf = tag == "bf16"
    if bf:
        cond, unc = [torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16) for x in (cond, unc)]
    out = oracle.cfg_mask_topk(cond, unc, 3.0, model=oracle.MODEL_LUMINA, pos_ids=d["o7.pos"], pos_base=int(d["o7.img_start"]) + 3,
                               top_k=2000, w=48, h=48, img_lo=m["img_lo"], img_hi=m["img_hi"], newline_id=m["syntax"][2],
                               eos_id=m["syntax"][0], bf16=bf)
    o7_full_check(out, d, tag)


def kv_full_data():
    d = H.load("evaluate_posterior_full.npz")
    slab = CS.kv_full_inputs()
    assert hashlib.sha256(slab.tobytes()).hexdigest() == str(d["kv.before_sha256"])
    return d, slab


def test_kv_and_hidden_gather_at_the_7b_slab_geometry():
    """O9 + O10 on the reference's own [64, 2, 32, S, 128] slab (kv_cache.py:101-122) with the default tree's retrieve rows."""
    d, slab = kv_full_data()
    best, alen, prev = int(d["kv.best"]), int(d["kv.accept_len"]), int(d["kv.prev"])
    row = d["kv.retrieve"][best]
    oracle.kv_gather(slab, row, alen + 1, prev)
    assert hashlib.sha256(slab.tobytes()).hexdigest() == str(d["kv.after_sha256"])
    assert np.all(d["kv.current_length"] == prev + alen + 1)
    assert np.array_equal(oracle.hidden_gather(d["kv.hidden"], row, alen + 1), d["kv.accept_hidden"])
    assert np.array_equal(d["kv.cand"][best, :alen + 1], d["kv.new_ids_tail"])
    p = np.zeros(65536, np.float32)
    p[4321] = 1.0
    assert oracle.sample_inverse_cdf(p, 0.77) == int(d["kv.token"].reshape(-1)[0]) == 4321
