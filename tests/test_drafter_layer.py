"""The drafter's decoder layer (lantern_amd/drafters/decoder_layer.py) against vectors recorded from the reference's
ChameleonDecoderLayer (tests/golden/layer.npz, make_golden_layer.py): the torch path in f32 on the CPU (host logic: norms, rotary, cache
concat, mask, GQA) and -- GPU -- the HIP skinny-GEMM path in bf16 within bf16 tolerance."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lantern_amd.drafters.decoder_layer import DecoderLayer  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "layer.npz"))
CASES = sorted({k.split(".")[0] for k in GOLD.files})


def build(name, device, dtype):
    hidden, heads, kv_heads, inter, mp, seed = (int(x) for x in GOLD[name + ".cfg"])
    cfg = types.SimpleNamespace(hidden_size=hidden, intermediate_size=inter, num_attention_heads=heads, num_key_value_heads=kv_heads,
                                max_position_embeddings=256, model_parallel_size=mp, rope_theta=10000.0, rms_norm_eps=1e-5, attention_bias=False,
                                mlp_bias=False, hidden_act="silu")
    layer = DecoderLayer(cfg, 0).eval()
    # the generator's recipe (make_golden_layer.fill_parameters): numpy RandomState(seed), the reference's state_dict order and names
    names = [str(n) for n in GOLD[name + ".names"]]
    mine = layer.state_dict()
    assert set(names) == set(mine.keys()), (set(names) ^ set(mine.keys()))          # the reference's parameter names load as they are
    rs = np.random.RandomState(seed)
    sd = {}
    for n in names:
        a = rs.standard_normal(tuple(mine[n].shape)).astype(np.float32)
        if mine[n].dim() > 1 and "norm" not in n:
            a = a / np.sqrt(mine[n].shape[-1])
        elif n.endswith("weight"):
            a = 1.0 + 0.1 * a
        else:
            a = 0.1 * a
        sd[n] = torch.from_numpy(a)
    layer.load_state_dict(sd, strict=True)
    return layer.to(device=device, dtype=dtype)


def g(name, key, device, dtype=None):
    t = torch.from_numpy(GOLD[name + "." + key]).to(device)
    return t.to(dtype) if (dtype is not None and t.is_floating_point()) else t


@pytest.mark.parametrize("name", CASES)
def test_layer_torch_path_matches_the_reference_layer_f32(name):
    layer = build(name, "cpu", torch.float32)
    with torch.no_grad():
        y0, kv0 = layer(g(name, "x0", "cpu"), attention_mask=g(name, "m0", "cpu"), position_ids=g(name, "pos0", "cpu"), use_cache=True)
        y1, kv1 = layer(g(name, "x1", "cpu"), attention_mask=g(name, "m1", "cpu"), position_ids=g(name, "pos1", "cpu"), past_key_value=kv0, use_cache=True)
    for got, key in ((y0, "y0"), (kv0[0], "k0"), (kv0[1], "v0"), (y1, "y1"), (kv1[0], "k1"), (kv1[1], "v1")):
        np.testing.assert_allclose(got.numpy(), GOLD[name + "." + key], rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per_projection"])
@pytest.mark.parametrize("name", CASES)
def test_layer_hip_path_matches_the_reference_layer_bf16(name, fused, monkeypatch):
    """Tolerance: bf16 has 8 bits of mantissa and every projection rounds its output to bf16 (as torch's bf16 nn.Linear does): the layer's
    output is compared with the reference's f32 result at 4e-2 absolute + 4e-2 relative, and with the SAME layer on torch's own bf16
    ops at 2e-2 (two bf16 pipelines that differ in accumulation order).  fused: rmsnorm / fused qkv / head-norm + rotary / o_proj + residual /
    gate-up with silu * up / down + residual kernels, the GEMMs in stream-K form (head_dim 64 or 128); per_projection: the skinny GEMM under torch's element-wise ops."""
    from lantern_amd import ops
    calls = []
    for fn in ("linear_rows", "linear_rows_streamk", "rmsnorm_rows", "qk_norm_rope"):
        real = getattr(ops, fn)
        monkeypatch.setattr(ops, fn, (lambda real, fn: (lambda *a, **kw: (calls.append(fn), real(*a, **kw))[1]))(real, fn))
    dev = torch.device("cuda")
    layer = build(name, dev, torch.bfloat16)
    layer.fused = fused
    bf = torch.bfloat16
    if fused and layer.self_attn.head_dim not in (64, 128):
        # outside the fused kernels the drafting shape raises instead of sliding onto torch's ops unnoticed
        from lantern_amd._lib import LanternError
        with pytest.raises(LanternError, match="fused HIP path"):
            layer(g(name, "x0", dev, bf), attention_mask=g(name, "m0", dev), position_ids=g(name, "pos0", dev), use_cache=True)
        return
    with torch.no_grad():
        y0, kv0 = layer(g(name, "x0", dev, bf), attention_mask=g(name, "m0", dev), position_ids=g(name, "pos0", dev), use_cache=True)
        y1, kv1 = layer(g(name, "x1", dev, bf), attention_mask=g(name, "m1", dev), position_ids=g(name, "pos1", dev), past_key_value=kv0, use_cache=True)
    head_dim = layer.self_attn.head_dim
    if fused and head_dim in (64, 128):
        assert (calls.count("qk_norm_rope") == 2 and calls.count("rmsnorm_rows") == 4 and calls.count("linear_rows_streamk") == 8 and
                calls.count("linear_rows") == 0), calls
    else:
        assert calls.count("linear_rows") == 8 and "qk_norm_rope" not in calls, calls          # fused qkv, o_proj, fused gate/up, down_proj per call
    for got, key in ((y0, "y0"), (y1, "y1"), (kv1[0], "k1"), (kv1[1], "v1")):
        np.testing.assert_allclose(got.float().cpu().numpy(), GOLD[name + "." + key], rtol=4e-2, atol=4e-2)
    # the same layer through torch's bf16 ops only (the kernels switched off)
    monkeypatch.setattr("lantern_amd.drafters.decoder_layer._hip_ok", lambda x, w: False)
    with torch.no_grad():
        t0, tkv0 = layer(g(name, "x0", dev, bf), attention_mask=g(name, "m0", dev), position_ids=g(name, "pos0", dev), use_cache=True)
        t1, _ = layer(g(name, "x1", dev, bf), attention_mask=g(name, "m1", dev), position_ids=g(name, "pos1", dev), past_key_value=tkv0, use_cache=True)
    np.testing.assert_allclose(y0.float().cpu().numpy(), t0.float().cpu().numpy(), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(y1.float().cpu().numpy(), t1.float().cpu().numpy(), rtol=2e-2, atol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", [c for c in CASES if c != "mha"])          # (mha: head_dim 16, outside the fused kernels)
def test_layer_tree_attention_path_matches_the_reference_layer(name, monkeypatch):
    """The drafting call with the tree block handed over as ancestor words (tree_bits / kv_start: what cnets.Model.forward passes) runs its
    attention on lantern_tree_attention -- no torch attention call -- and reproduces the reference layer's decode output (the vectors recorded
    from ChameleonDecoderLayer with the additive tree mask) within the bf16 tolerance of the fused path."""
    import torch.nn.functional as F
    from lantern_amd import ops
    dev, bf = torch.device("cuda"), torch.bfloat16
    layer = build(name, dev, bf)
    if layer.self_attn.head_dim not in (64, 128):
        pytest.skip("head_dim outside the fused kernels")
    calls = []
    real_ta, real_sdpa = ops.tree_attention, F.scaled_dot_product_attention
    monkeypatch.setattr(ops, "tree_attention", lambda *a, **kw: (calls.append("tree"), real_ta(*a, **kw))[1])
    monkeypatch.setattr(F, "scaled_dot_product_attention", lambda *a, **kw: (calls.append("sdpa"), real_sdpa(*a, **kw))[1])
    m1 = g(name, "m1", dev)
    T1 = m1.shape[2]
    tree = (m1[0, 0, :, -T1:] == 0).float()                       # the tree block of the recorded mask (row 0: no padding)
    bits, t1 = ops.drafter_tree_bits(tree[None, None], T1)
    start = (m1[:, 0, -1, :] == 0).to(torch.int64).argmax(dim=1)  # first visible key per row (row 1 is left-padded by 2)
    with torch.no_grad():
        y0, kv0 = layer(g(name, "x0", dev, bf), attention_mask=g(name, "m0", dev), position_ids=g(name, "pos0", dev), use_cache=True)
        calls.clear()
        y1, kv1 = layer(g(name, "x1", dev, bf), attention_mask=m1, position_ids=g(name, "pos1", dev), past_key_value=kv0, use_cache=True,
                        tree_bits=bits, tree_keys=t1, kv_start=start)
    assert calls == ["tree"], calls
    for got, key in ((y1, "y1"), (kv1[0], "k1"), (kv1[1], "v1")):
        np.testing.assert_allclose(got.float().cpu().numpy(), GOLD[name + "." + key], rtol=4e-2, atol=4e-2)
    # and the additive-mask form of the same call agrees -- restated as ancestor words on the way in (additive_mask_words), so it is the SAME kernel
    # and the same bits: never torch's scaled_dot_product_attention on device tensors at the HIP shapes
    calls.clear()
    with torch.no_grad():
        y1m, _ = layer(g(name, "x1", dev, bf), attention_mask=m1, position_ids=g(name, "pos1", dev), past_key_value=kv0, use_cache=True)
    assert calls == ["tree"], calls
    assert torch.equal(y1, y1m)
    # a mask that is not (left padding) x (a block over the new tokens) has no HIP form: refused, not routed to torch
    from lantern_amd._lib import LanternError
    holey = m1.clone()
    holey[:, :, 0, 1] = torch.finfo(holey.dtype).min
    holey[:, :, 0, 0] = 0
    with pytest.raises(LanternError, match="no HIP kernel"):
        layer(g(name, "x1", dev, bf), attention_mask=holey, position_ids=g(name, "pos1", dev), past_key_value=kv0, use_cache=True)
    assert "sdpa" not in calls


@pytest.mark.gpu
def test_layer_inplace_cache_equals_the_concatenated_cache():
    """inplace_cache: `present` is a growing view of a layer-owned slab.  The drafter's pattern -- prefix forward, several depth calls that
    each extend the previous present, then a new cycle from the SHORTER prefix cache -- gives bit-identical outputs to the torch.cat cache,
    through a slab reallocation (capacity 256 -> 512) and an external (non-slab) cache handed in."""
    name = "hd64_gqa"
    dev, bf = torch.device("cuda"), torch.bfloat16
    outs = []
    for inplace in (False, True):
        torch.manual_seed(3)
        layer = build(name, dev, bf)
        layer.inplace_cache = inplace
        H = layer.hidden_size
        res = []

        def call(T, past, pos0):
            x = torch.randn(2, T, H, device=dev, dtype=bf)
            L = 0 if past is None else past[0].shape[2]
            pos = (pos0 + torch.arange(T, device=dev))[None].expand(2, T).contiguous()
            m = torch.zeros(2, 1, T, L + T, device=dev)
            m[:, :, :, L:] = torch.full((T, T), torch.finfo(torch.float32).min, device=dev).triu(1)
            with torch.no_grad():
                y, kv = layer(x, attention_mask=m, position_ids=pos, past_key_value=past, use_cache=True)
            res.append(y.clone())
            return kv
        stable = call(30, None, 0)                      # prefix
        kv = stable
        for d in range(4):                              # depth calls extend the previous present
            kv = call(10, kv, 30 + d)
        res.append(kv[0].clone()); res.append(kv[1].clone())
        stable2 = call(3, stable, 30)                   # next cycle: from the shorter prefix (tree rows are overwritten)
        kv = stable2
        for d in range(25):                             # grows past 256 rows: the slab is reallocated
            kv = call(10, kv, 33 + d)
        ext = (kv[0].clone(), kv[1].clone())            # a cache that is not the slab
        call(5, ext, 283)
        res.append(kv[0].clone())
        outs.append(res)
    assert len(outs[0]) == len(outs[1])
    for i, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,epi", [(20, 4096, 12288, 0), (20, 4096, 4096, 1), (20, 4096, 11008, 2), (20, 11008, 4096, 1), (1, 512, 96, 0),
                                       (32, 272, 40, 1), (7, 1040, 1000, 2), (32, 16, 33, 0), (9, 832, 70, 2), (32, 64, 1, 1)])
def test_streamk_gemm_matches_f64_and_the_per_tile_kernels(M, K, N, epi):
    """lantern_linear_rows_streamk (one launch, equal weight shares per workgroup, partial tiles met in the workspace) against the exact
    product in f64 (tolerance: bf16 inputs, f32 accumulation over K in a different order -- 2e-2 of the row scale) and against the per-tile
    kernels it replaces in the layer (same roundings in the epilogue: within one bf16 ulp of each other); the drafting shapes of the 7B layer,
    ragged tiles / K not a multiple of the chunk, one row; twice in a row on the same workspace (the counters come back to zero)."""
    from lantern_amd import ops
    dev, bf = torch.device("cuda"), torch.bfloat16
    gen = torch.Generator(device="cuda").manual_seed(M * 131 + K + N + epi)
    x = torch.randn(M, K, device=dev, dtype=bf, generator=gen)
    rows = 2 * N if epi == 2 else N
    w = (torch.randn(rows, K, device=dev, generator=gen) / K ** 0.5).to(bf)
    b = (0.1 * torch.randn(rows, device=dev, generator=gen)).to(bf)
    res = torch.randn(M, N, device=dev, dtype=bf, generator=gen)
    xd, wd, bd = x.double(), w.double(), b.double()
    if epi == 0:
        want = xd @ wd.T + bd
        old = ops.linear_rows(x, w, 0, N, bias=b)
        kw = {}
    elif epi == 1:
        want = (xd @ wd.T + bd) + res.double()
        old = ops.linear_rows_epilogue(x, w, ops.EPI_RESIDUAL, bias=b, residual=res)
        kw = dict(residual=res)
    else:
        gate, up = xd @ wd[:N].T + bd[:N], xd @ wd[N:].T + bd[N:]
        want = torch.nn.functional.silu(gate) * up
        old = ops.linear_rows_epilogue(x, w, ops.EPI_SILU_MUL, bias=b, pair_rows=N)
        kw = dict(pair_rows=N)
    forms = [w]
    if K % 64 == 0:                          # the brick layout (what the layer streams): same values, same contraction, another address order
        forms.append(ops.pack_linear_weight(w, N if epi == 2 else 0))
    for wt in forms:
        for _ in range(2):
            got = ops.linear_rows_streamk(x, wt, epi, bias=b, **kw)
            assert got.shape == (M, N)
            scale = want.abs().max().item()
            assert (got.double() - want).abs().max().item() <= 2e-2 * scale
            assert (got.float() - old.float()).abs().max().item() <= 2e-2 * scale
        again = ops.linear_rows_streamk(x, wt, epi, bias=b, **kw)
        assert torch.equal(again, got)          # deterministic: the partials are added in K order whoever finishes a tile


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,epi", [(20, 16384, 64, 0), (32, 8192, 96, 1), (20, 11008, 40, 2), (8, 4096, 32, 0)])
def test_streamk_split_tiles_survive_a_poisoned_workspace(M, K, N, epi):
    """Few tiles, many workgroups: every tile is split over dozens of workgroups on different XCDs, so each output element is the sum of partial
    tiles that met in the workspace.  Between launches the partial-tile region is filled with NaN (the tile counters behind it stay zero, as the
    kernel leaves them): a finisher that read a slot before its writer's stores had landed would return NaN or a stale sum.  The publish
    protocol (each thread waits for its own device-coherent stores -- s_waitcnt vmcnt(0) -- before the barrier that precedes the counter add)
    must give the same bits on every launch."""
    from lantern_amd import ops, _lib
    dev, bf = torch.device("cuda"), torch.bfloat16
    gen = torch.Generator(device="cuda").manual_seed(7 * M + K + N + epi)
    x = torch.randn(M, K, device=dev, dtype=bf, generator=gen)
    rows = 2 * N if epi == 2 else N
    w = (torch.randn(rows, K, device=dev, generator=gen) / K ** 0.5).to(bf)
    res = torch.randn(M, N, device=dev, dtype=bf, generator=gen)
    kw = dict(residual=res) if epi == 1 else (dict(pair_rows=N) if epi == 2 else {})
    xd, wd = x.double(), w.double()
    want = xd @ wd[:N].T
    if epi == 1:
        want = want + res.double()
    if epi == 2:
        want = torch.nn.functional.silu(want) * (xd @ wd[N:].T)
    wt = ops.pack_linear_weight(w, N if epi == 2 else 0) if K % 64 == 0 else w
    first = ops.linear_rows_streamk(x, wt, epi, **kw)
    assert (first.double() - want).abs().max().item() <= 2e-2 * want.abs().max().item()
    ws = ops._sk_workspace(dev)
    L = _lib.lib()
    partial_bytes = int(L.lantern_linear_rows_streamk_workspace(32)) - 4 - 256          # the partial tiles come first, the counters behind them
    assert partial_bytes > 0 and partial_bytes % 4 == 0
    partials = ws[:partial_bytes].view(torch.float32)
    counters = ws[partial_bytes:]
    for _ in range(60):
        partials.fill_(float("nan"))
        got = ops.linear_rows_streamk(x, wt, epi, **kw)
        assert torch.equal(got, first)
    torch.cuda.synchronize()
    assert int(counters.to(torch.int64).sum().item()) == 0          # every tile counter back at zero


# ----------------------------------------------------------------------------- LlamaGen's and Anole's drafter layers (tests/golden/layer_lg.npz)
GOLD_LG = np.load(os.path.join(ROOT, "tests", "golden", "layer_lg.npz"))
LG_CASES = sorted({k.split(".")[0] for k in GOLD_LG.files if k.startswith("lg_") and "." in k})
AN_CASES = sorted({k.split(".")[0] for k in GOLD_LG.files if k.startswith("an_")})


def _fill(layer, names, seed):
    """make_golden_layer.fill_parameters: numpy RandomState(seed) in the reference's state_dict order."""
    mine = layer.state_dict()
    assert set(names) == set(mine.keys()), (set(names) ^ set(mine.keys()))          # the reference's parameter names load as they are
    rs = np.random.RandomState(seed)
    sd = {}
    for n in names:
        a = rs.standard_normal(tuple(mine[n].shape)).astype(np.float32)
        if mine[n].dim() > 1 and "norm" not in n:
            a = a / np.sqrt(mine[n].shape[-1])
        elif n.endswith("weight"):
            a = 1.0 + 0.1 * a
        else:
            a = 0.1 * a
        sd[n] = torch.from_numpy(a)
    layer.load_state_dict(sd, strict=True)


def build_lg(name, device, dtype):
    from lantern_amd.drafters.decoder_layer import LlamaDecoderLayer
    hidden, heads, kv_heads, inter, index, seed = (int(x) for x in GOLD_LG[name + ".cfg"])
    cfg = types.SimpleNamespace(hidden_size=hidden, intermediate_size=inter, num_attention_heads=heads, num_key_value_heads=kv_heads, rms_norm_eps=1e-6,
                                hidden_act="silu")
    layer = LlamaDecoderLayer(cfg, index).eval()
    _fill(layer, [str(n) for n in GOLD_LG[name + ".names"]], seed)
    return layer.to(device=device, dtype=dtype)


def build_an(name, device, dtype):
    """Anole's layer = the Chameleon layer with one head-norm row per HEAD (cnets_anole.py:317-332): DecoderLayer with model_parallel_size = heads."""
    hidden, heads, kv_heads, inter, eps9, seed = (int(x) for x in GOLD_LG[name + ".cfg"])
    cfg = types.SimpleNamespace(hidden_size=hidden, intermediate_size=inter, num_attention_heads=heads, num_key_value_heads=kv_heads,
                                max_position_embeddings=256, model_parallel_size=heads, rope_theta=10000.0, rms_norm_eps=eps9 * 1e-9, attention_bias=False,
                                mlp_bias=False, hidden_act="silu")
    layer = DecoderLayer(cfg, 0).eval()
    _fill(layer, [str(n) for n in GOLD_LG[name + ".names"]], seed)
    return layer.to(device=device, dtype=dtype)


def gl(name, key, device, dtype=None):
    t = torch.from_numpy(GOLD_LG[name + "." + key]).to(device)
    return t.to(dtype) if (dtype is not None and t.is_floating_point()) else t


def test_llamagen_freqs_table_equals_the_reference_table():
    from lantern_amd.drafters.decoder_layer import precompute_freqs_cis_2d
    assert np.array_equal(precompute_freqs_cis_2d(16, 64, 10000, 119).numpy(), GOLD_LG["lg_table_16_64_119"])
    assert np.array_equal(precompute_freqs_cis_2d(24, 64, 10000, 0).numpy(), GOLD_LG["lg_table_24_64_0"])


@pytest.mark.parametrize("name", LG_CASES)
def test_llamagen_layer_torch_path_matches_the_reference_layer_f32(name):
    layer = build_lg(name, "cpu", torch.float32)
    with torch.no_grad():
        y0, kv0 = layer(gl(name, "x0", "cpu"), attention_mask=gl(name, "m0", "cpu"), position_ids=gl(name, "pos0", "cpu"), freqs_cis=gl(name, "f0", "cpu"), use_cache=True)
        y1, kv1 = layer(gl(name, "x1", "cpu"), attention_mask=gl(name, "m1", "cpu"), position_ids=gl(name, "pos1", "cpu"), freqs_cis=gl(name, "f1", "cpu"),
                        past_key_value=kv0, use_cache=True)
    for got, key in ((y0, "y0"), (kv0[0], "k0"), (kv0[1], "v0"), (y1, "y1"), (kv1[0], "k1"), (kv1[1], "v1")):
        np.testing.assert_allclose(got.numpy(), GOLD_LG[name + "." + key], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", AN_CASES)
def test_anole_layer_torch_path_matches_the_reference_layer_f32(name):
    layer = build_an(name, "cpu", torch.float32)
    with torch.no_grad():
        y0, kv0 = layer(gl(name, "x0", "cpu"), attention_mask=gl(name, "m0", "cpu"), position_ids=gl(name, "pos0", "cpu"), use_cache=True)
        y1, kv1 = layer(gl(name, "x1", "cpu"), attention_mask=gl(name, "m1", "cpu"), position_ids=gl(name, "pos1", "cpu"), past_key_value=kv0, use_cache=True)
    for got, key in ((y0, "y0"), (kv0[0], "k0"), (kv0[1], "v0"), (y1, "y1"), (kv1[0], "k1"), (kv1[1], "v1")):
        np.testing.assert_allclose(got.numpy(), GOLD_LG[name + "." + key], rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("tree", [False, True], ids=["additive_mask", "tree_bits"])
@pytest.mark.parametrize("name", LG_CASES + AN_CASES)
def test_llamagen_and_anole_layers_hip_path_match_the_reference_layers_bf16(name, tree, monkeypatch):
    """The fused HIP path (stream-K GEMMs, rmsnorm_rows, the head stage -- lantern_qk_rope_pairs for LlamaGen, lantern_qk_norm_rope with per-head
    norm rows for Anole --, tree attention when the tree block comes as ancestor words) against the vectors recorded from the reference's
    LlamaDecoderLayer / Anole ChameleonDecoderLayer: bf16 tolerance as for the Lumina layer (4e-2 + 4e-2 relative against the f32 reference)."""
    import torch.nn.functional as F
    from lantern_amd import ops
    from lantern_amd._lib import LanternError
    dev, bf = torch.device("cuda"), torch.bfloat16
    lg = name.startswith("lg_")
    layer = (build_lg if lg else build_an)(name, dev, bf)
    kw0 = dict(freqs_cis=gl(name, "f0", dev)) if lg else {}
    kw1 = dict(freqs_cis=gl(name, "f1", dev)) if lg else {}
    if layer.self_attn.head_dim not in (64, 128):
        with pytest.raises(LanternError, match="fused HIP path"):
            layer(gl(name, "x0", dev, bf), attention_mask=gl(name, "m0", dev), position_ids=gl(name, "pos0", dev), use_cache=True, **kw0)
        return
    calls = []
    for fn in ("linear_rows_streamk", "rmsnorm_rows", "qk_norm_rope", "qk_rope_pairs", "tree_attention"):
        real = getattr(ops, fn)
        monkeypatch.setattr(ops, fn, (lambda real, fn: (lambda *a, **k_: (calls.append(fn), real(*a, **k_))[1]))(real, fn))
    real_sdpa = F.scaled_dot_product_attention
    monkeypatch.setattr(F, "scaled_dot_product_attention", lambda *a, **k_: (calls.append("sdpa"), real_sdpa(*a, **k_))[1])
    m1 = gl(name, "m1", dev)
    if tree:
        T1 = m1.shape[2]
        bits, t1 = ops.drafter_tree_bits((m1[0, 0, :, -T1:] == 0).float()[None, None], T1)
        start = (m1[:, 0, -1, :] == 0).to(torch.int64).argmax(dim=1)
        kw1 = dict(kw1, tree_bits=bits, tree_keys=t1, kv_start=start)
    with torch.no_grad():
        y0, kv0 = layer(gl(name, "x0", dev, bf), attention_mask=gl(name, "m0", dev), position_ids=gl(name, "pos0", dev), use_cache=True, **kw0)
        n0 = len(calls)
        y1, kv1 = layer(gl(name, "x1", dev, bf), attention_mask=m1, position_ids=gl(name, "pos1", dev), past_key_value=kv0, use_cache=True, **kw1)
    second = calls[n0:]
    assert second.count("linear_rows_streamk") == 4 and second.count("qk_rope_pairs" if lg else "qk_norm_rope") == 1, second
    # (round 5: the additive-mask form is restated as ancestor words + kv_start on the way in -- the same HIP attention, never torch's SDPA)
    assert second.count("tree_attention") == 1 and second.count("sdpa") == 0, second
    n_norm = 2 if not lg else (1 if int(GOLD_LG[name + ".cfg"][4]) == 0 else 2)          # LlamaGen's layer 0 has no input norm (EAGLE)
    assert second.count("rmsnorm_rows") == n_norm, second
    for got, key in ((y0, "y0"), (y1, "y1"), (kv1[0], "k1"), (kv1[1], "v1")):
        np.testing.assert_allclose(got.float().cpu().numpy(), GOLD_LG[name + "." + key], rtol=4e-2, atol=4e-2)


# ----------------------------------------------------------------------------- prefills: any number of rows on the packed weights
@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,epi", [(300, 4096, 4096, 0), (129, 1024, 352, 1), (1200, 1280, 3584, 2), (33, 64, 40, 0), (65, 11008, 96, 1), (64, 256, 64, 2),
                                       (20, 512, 70, 2), (1, 64, 1, 0), (130, 64, 40, 0), (257, 128, 33, 1), (131, 192, 40, 2), (513, 64, 1, 0), (129, 64, 129, 2)])
def test_packed_gemm_of_any_row_count_matches_f64_and_the_streamk_kernel(M, K, N, epi):
    """lantern_linear_rows_packed (row blocks of 128 / 64 over the packed bricks: the prompt prefill of the drafter's layer) against the exact
    product in f64 (2e-2 of the row scale, as the stream-K test) and -- where both apply, <= 32 rows -- bit-identical roundings to within one bf16
    ulp of lantern_linear_rows_streamk; ragged row blocks and column tiles."""
    from lantern_amd import ops
    dev, bf = torch.device("cuda"), torch.bfloat16
    gen = torch.Generator(device="cuda").manual_seed(M * 7 + K + N + epi)
    x = torch.randn(M, K, device=dev, dtype=bf, generator=gen)
    rows = 2 * N if epi == 2 else N
    w = (torch.randn(rows, K, device=dev, generator=gen) / K ** 0.5).to(bf)
    b = (0.1 * torch.randn(rows, device=dev, generator=gen)).to(bf)
    res = torch.randn(M, N, device=dev, dtype=bf, generator=gen)
    xd, wd, bd = x.double(), w.double(), b.double()
    if epi == 0:
        want, kw = xd @ wd.T + bd, {}
    elif epi == 1:
        want, kw = (xd @ wd.T + bd) + res.double(), dict(residual=res)
    else:
        want, kw = torch.nn.functional.silu(xd @ wd[:N].T + bd[:N]) * (xd @ wd[N:].T + bd[N:]), {}
    pk = ops.pack_linear_weight(w, N if epi == 2 else 0)
    got = ops.linear_rows_packed(x, pk, epi, bias=b, **kw)
    assert got.shape == (M, N)
    scale = want.abs().max().item()
    assert (got.double() - want).abs().max().item() <= 2e-2 * scale
    assert torch.equal(ops.linear_rows_packed(x, pk, epi, bias=b, **kw), got)
    if M <= 32:
        sk = ops.linear_rows_streamk(x, pk, epi, bias=b, **dict(kw, **(dict(pair_rows=N) if epi == 2 else {})))
        assert (got.float() - sk.float()).abs().max().item() <= 2e-2 * scale


def _causal_additive(T0, past, starts, device):
    """[B, 1, T0, past + T0] additive mask: causal among the new tokens, keys in front of starts[b] hidden, the diagonal always open."""
    fmin = torch.finfo(torch.float32).min
    B, S = len(starts), past + T0
    q = torch.arange(T0, device=device)[:, None] + past
    k = torch.arange(S, device=device)[None]
    m = torch.zeros(B, 1, T0, S, device=device)
    for b, st in enumerate(starts):
        vis = (k <= q) & ((k >= st) | (k == q))
        m[b, 0] = torch.where(vis, 0.0, fmin)
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("kind,name", [("lumina", "hd128"), ("lumina", "hd64_gqa"), ("anole", "an_hd64"), ("anole", "an_hd128"), ("llamagen", "lg_hd64"), ("llamagen", "lg_hd64_idx1"),
                                       ("llamagen", "lg_hd128")])
def test_layer_prefill_runs_on_the_hip_path_and_matches_the_torch_path(kind, name, monkeypatch):
    """A prompt prefill (2 x 150 rows, the second row left-padded by 5, then 2 x 70 more rows behind that cache): with the `causal` hint of
    cnets.Model.forward the layer stays on the HIP kernels -- lantern_linear_rows_packed for the four GEMMs, the head stage, lantern_tree_attention
    block by block -- and agrees with the same layer's torch path in f32 under the equivalent additive mask (4e-2 + 4e-2 relative, the layer tests'
    bf16 tolerance) on every row that is not padding; the caches agree too."""
    import torch.nn.functional as F
    from lantern_amd import ops
    from lantern_amd.drafters.decoder_layer import precompute_freqs_cis_2d
    dev, bf = torch.device("cuda"), torch.bfloat16
    mk = {"lumina": build, "anole": build_an, "llamagen": build_lg}[kind]
    hip, ref = mk(name, dev, bf), mk(name, dev, torch.float32)
    ref.fused = False
    H, d = hip.self_attn.hidden_size, hip.self_attn.head_dim
    T0, T1, starts = 150, 70, [0, 5]
    gen = torch.Generator(device="cuda").manual_seed(len(name) + T0)
    x0, x1 = (torch.randn(2, T, H, device=dev, generator=gen).to(bf) for T in (T0, T1))
    if kind == "llamagen":
        table = precompute_freqs_cis_2d(16, d, 10000, 20).to(dev)
        pos0, pos1 = torch.arange(T0, device=dev)[None], torch.arange(T0, T0 + T1, device=dev)[None]
        kw0, kw1 = dict(freqs_cis=table[pos0[0]]), dict(freqs_cis=table[pos1[0]])
    else:
        st = torch.tensor(starts, device=dev)[:, None]
        pos0 = (torch.arange(T0, device=dev)[None] - st).clamp(min=0)
        pos1 = torch.arange(T0, T0 + T1, device=dev)[None] - st
        kw0 = kw1 = {}
    calls = []
    for fn in ("linear_rows_packed", "linear_rows_streamk", "tree_attention"):
        real = getattr(ops, fn)
        monkeypatch.setattr(ops, fn, (lambda real, fn: (lambda *a, **k_: (calls.append(fn), real(*a, **k_))[1]))(real, fn))
    real_sdpa = F.scaled_dot_product_attention
    monkeypatch.setattr(F, "scaled_dot_product_attention", lambda *a, **k_: (calls.append("sdpa"), real_sdpa(*a, **k_))[1])
    real_lin = F.linear
    monkeypatch.setattr(F, "linear", lambda *a, **k_: (calls.append("F.linear"), real_lin(*a, **k_))[1])
    ks = torch.tensor(starts, dtype=torch.int64, device=dev)
    m0, m1 = _causal_additive(T0, 0, starts, dev), _causal_additive(T1, T0, starts, dev)
    with torch.no_grad():
        y0, kv0 = hip(x0, attention_mask=m0, position_ids=pos0, use_cache=True, causal=True, kv_start=ks, **kw0)
        y1, kv1 = hip(x1, attention_mask=m1, position_ids=pos1, past_key_value=kv0, use_cache=True, causal=True, kv_start=ks, **kw1)
        assert calls.count("linear_rows_packed") == 8 and calls.count("tree_attention") == 3 + 2 and not {"sdpa", "F.linear", "linear_rows_streamk"} & set(calls), calls
        calls.clear()
        r0, rkv0 = ref(x0.float(), attention_mask=m0, position_ids=pos0, use_cache=True, **kw0)
        r1, rkv1 = ref(x1.float(), attention_mask=m1, position_ids=pos1, past_key_value=rkv0, use_cache=True, **kw1)
    for b, st in enumerate(starts):
        np.testing.assert_allclose(y0[b, st:].float().cpu().numpy(), r0[b, st:].cpu().numpy(), rtol=4e-2, atol=4e-2)
        np.testing.assert_allclose(y1[b].float().cpu().numpy(), r1[b].cpu().numpy(), rtol=4e-2, atol=4e-2)
        for got, want in zip(kv1, rkv1):
            np.testing.assert_allclose(got[b, :, st:].float().cpu().numpy(), want[b, :, st:].cpu().numpy(), rtol=4e-2, atol=4e-2)
