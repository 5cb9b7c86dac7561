"""`lantern_tree_attention` against a plain torch restatement of the reference's eager tree attention
(models/kv_variants/modeling_lumina_mgpt_kv.py:433-442 with the additive mask of :1508-1546): scores and softmax in f32,
probabilities rounded to bf16, then the V product.  Floating point, so the bar is a tolerance: the kernel keeps unnormalised
bf16 exponentials (flash style) where the reference rounds normalised probabilities, both accumulate in f32 and round the
output to bf16 -> |diff| <= 2e-2 + 2e-2*|ref| on outputs of O(1) magnitude (bf16 has 8 mantissa bits: 1 ulp = 0.8 % )."""
import math

import pytest
import torch

from lantern_amd import ops
from lantern_amd.drafters.choices import mc_sim_7b_63, naive_extend_57
from lantern_amd.verify import generate_tree_buffers

pytestmark = pytest.mark.gpu
ATOL, RTOL = 2e-2, 2e-2


def reference_attention(q, k_cache, v_cache, tree_mask, kv_len, kv_start, scale):
    """q [B,N,Hq,d]; caches [B,Hkv,S,d]; tree_mask [N,N] or [B,N,N] (non-zero = visible)."""
    B, N, Hq, d = q.shape
    Hkv = k_cache.shape[1]
    outs = []
    fmin = torch.finfo(torch.float32).min
    for b in range(B):
        L, st = int(kv_len[b]), int(kv_start[b])
        k = k_cache[b, :, :L].repeat_interleave(Hq // Hkv, dim=0)          # repeat_kv
        v = v_cache[b, :, :L].repeat_interleave(Hq // Hkv, dim=0)
        w = torch.matmul(q[b].transpose(0, 1), k.transpose(1, 2)).float() * scale   # bf16 matmul like the reference, then f32
        mask = torch.zeros(N, L, dtype=torch.float32, device=q.device)
        mask[:, :st] = fmin
        tm = tree_mask if tree_mask.dim() == 2 else tree_mask[b]
        mask[:, L - N:][tm == 0] = fmin
        p = torch.softmax(w + mask, dim=-1).to(torch.bfloat16)
        outs.append(torch.matmul(p, v).transpose(0, 1).reshape(N, Hq * d))
    return torch.stack(outs)


def exact_attention(q, k_cache, v_cache, tree_mask, kv_len, kv_start, scale):
    """Same mathematics in f64 (what both implementations approximate)."""
    B, N, Hq, d = q.shape
    Hkv = k_cache.shape[1]
    outs = []
    for b in range(B):
        L, st = int(kv_len[b]), int(kv_start[b])
        k = k_cache[b, :, :L].repeat_interleave(Hq // Hkv, dim=0).double()
        v = v_cache[b, :, :L].repeat_interleave(Hq // Hkv, dim=0).double()
        w = torch.matmul(q[b].transpose(0, 1).double(), k.transpose(1, 2)) * scale
        vis = torch.ones(N, L, dtype=torch.bool, device=q.device)
        vis[:, :st] = False
        tm = tree_mask if tree_mask.dim() == 2 else tree_mask[b]
        vis[:, L - N:] = tm != 0
        p = torch.softmax(w.masked_fill(~vis, -math.inf), dim=-1)
        outs.append(torch.matmul(p, v).transpose(0, 1).reshape(N, Hq * d))
    return torch.stack(outs)


def make_case(B, Hq, Hkv, N, d, S_max, lens, starts, seed, tree_mask=None, q_scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    q = (q_scale * torch.randn(B, N, Hq, d, generator=g)).to(torch.bfloat16).cuda()
    k = torch.randn(B, Hkv, S_max, d, generator=g).to(torch.bfloat16).cuda()
    v = torch.randn(B, Hkv, S_max, d, generator=g).to(torch.bfloat16).cuda()
    for b in range(B):           # rows past the end hold junk that must never leak (a stale cache, here: huge values and NaN)
        k[b, :, lens[b]:] = 1e30
        v[b, :, lens[b]:] = float("nan")
    if tree_mask is None:
        tree_mask = torch.tril(torch.ones(N, N))
        tree_mask[torch.rand(N, N, generator=g) < 0.5] = 0
        tree_mask = torch.tril(tree_mask)
        tree_mask.fill_diagonal_(1)
        tree_mask[:, 0] = 1
    return q, k, v, tree_mask.cuda(), torch.tensor(lens, dtype=torch.int64).cuda(), torch.tensor(starts, dtype=torch.int64).cuda()


def check(q, k, v, tm, lens, starts, max_kv_len=None, vs_ref=True):
    d = q.shape[-1]
    scale = d ** -0.5
    bits = ops.tree_mask_bits(tm)
    out = ops.tree_attention(q, k, v, bits, kv_len=lens, kv_start=starts, max_kv_len=max_kv_len)
    torch.cuda.synchronize()
    ref = reference_attention(q, k, v, tm, lens, starts, scale).float()
    exact = exact_attention(q, k, v, tm, lens, starts, scale).float()
    got = out.float()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    if vs_ref:
        assert (err <= ATOL + RTOL * ref.abs()).all(), float(err.max())
    # and it is no further from the exact result than the reference's own bf16 pipeline (plus one bf16 ulp of slack)
    assert float((got - exact).abs().max()) <= float((ref - exact).abs().max()) + 2e-2
    return got


@pytest.mark.parametrize("name,choices", [("mc_sim_7b_63", mc_sim_7b_63), ("naive_extend_57", naive_extend_57)])
def test_static_trees_lumina_shapes(name, choices):
    """Lumina / Anole head shape (d = 128, MHA), the two static trees of the baseline configs, cond + uncond rows of different
    length, the uncond row behind left padding."""
    tb = generate_tree_buffers(choices, device="cuda")
    tm = tb["tree_attn_mask"][0, 0]
    N = tm.shape[0]
    case = make_case(2, 8, 8, N, 128, 640, [531 + N, 407 + N], [0, 37], seed=1, tree_mask=tm.cpu())
    check(*case)


@pytest.mark.parametrize("N", [1, 2, 26, 32, 33, 59, 64])
def test_tree_sizes(N):
    case = make_case(1, 4, 4, N, 128, 512, [300 + N], [0], seed=10 + N)
    check(*case)


@pytest.mark.parametrize("d,Hq,Hkv", [(64, 20, 20), (64, 8, 2), (128, 8, 2), (128, 6, 1)])
def test_head_shapes_and_gqa(d, Hq, Hkv):
    """LlamaGen's d = 64 with 20 heads; grouped KV heads (repeat_kv in the reference)."""
    case = make_case(2, Hq, Hkv, 59, d, 384, [200, 384], [0, 11], seed=3)
    check(*case)


@pytest.mark.parametrize("prev", [0, 1, 31, 32, 33, 95, 96])
def test_prefix_lengths_around_tile_edges(prev):
    """The tree block may start anywhere in a 32-key tile; prev = 0 is a forward with an empty cache."""
    N = 26
    case = make_case(1, 2, 2, N, 128, 256, [prev + N], [0], seed=20 + prev)
    check(*case)


@pytest.mark.parametrize("start", [0, 1, 31, 32, 40, 63, 64, 100])
def test_left_padding(start):
    case = make_case(2, 2, 2, 26, 128, 256, [100 + 26, 100 + 26], [start, 0], seed=30 + start)
    check(*case)


def test_per_row_tree_bits_and_transposed_q():
    B, Hq, N, d = 3, 4, 40, 128
    g = torch.Generator(device="cpu").manual_seed(5)
    tms = []
    for b in range(B):
        tm = torch.tril((torch.rand(N, N, generator=g) < 0.4).float())
        tm.fill_diagonal_(1)
        tms.append(tm)
    tm = torch.stack(tms)
    q, k, v, _, lens, starts = make_case(B, Hq, Hq, N, d, 320, [150, 320, 41], [0, 5, 0], seed=6)
    tm = tm.cuda()
    got = check(q, k, v, tm, lens, starts)
    qt = q.permute(0, 2, 1, 3).contiguous()                      # [B,Hq,N,d] as after .transpose(1, 2) in the reference
    out2 = ops.tree_attention(qt.permute(0, 2, 1, 3), k, v, ops.tree_mask_bits(tm), kv_len=lens, kv_start=starts)
    assert torch.equal(out2.float(), got)


def test_split_keys_path_matches_single_pass(monkeypatch):
    """Few (batch row, head) pairs and a long cache: the key range is split over gridDim.y and merged by the second kernel."""
    from lantern_amd import _lib
    case = make_case(1, 2, 2, 59, 128, 4096, [3900], [17], seed=8)
    q, k, v, tm, lens, starts = case
    need = _lib.lib().lantern_tree_attention_workspace(1, 2, 59, 128, 4096)
    assert need > 0                                              # this shape does take the split path
    got = check(*case)
    got_tight = check(q, k, v, tm, lens, starts, max_kv_len=3900)
    assert (got - got_tight).abs().max() <= 2e-2


def test_large_logits_do_not_overflow():
    """Scores of +-300: the softmax is nearly one-hot.  The reference rounds its scores to bf16 (a bf16 matmul output, +-1 at this
    magnitude) before the softmax, the kernel keeps them in f32, so here the yardstick is the f64 result: the kernel must be at
    least as close to it as the reference's own pipeline."""
    case = make_case(1, 2, 2, 26, 128, 256, [200], [0], seed=9, q_scale=30.0)
    check(*case, vs_ref=False)


def test_argument_errors():
    from lantern_amd._lib import LanternError
    q, k, v, tm, lens, starts = make_case(1, 2, 2, 8, 128, 64, [40], [0], seed=11)
    bits = ops.tree_mask_bits(tm)
    with pytest.raises(LanternError):
        ops.tree_attention(q.float(), k, v, bits)
    with pytest.raises(LanternError):
        ops.tree_attention(q, k, v[:, :, :32], bits)
    with pytest.raises(LanternError):
        ops.tree_attention(q, k, v, bits[:4])
    with pytest.raises(LanternError):
        ops.tree_attention(q, k, v, bits, max_kv_len=4)
    with pytest.raises(LanternError):
        ops.tree_mask_bits(torch.ones(65, 65))
    q3 = torch.zeros(1, 8, 2, 96, dtype=torch.bfloat16, device="cuda")
    k3 = torch.zeros(1, 2, 64, 96, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(LanternError, match="head_dim"):
        ops.tree_attention(q3, k3, k3.clone(), bits)


def test_drop_in_for_the_eager_block_on_kvcache_slabs():
    """The INTEGRATION.md stub end to end: K/V appended with the reference's `KVCache.cat` into the `[2L,B,Hkv,S_max,d]` slab of
    `initialize_past_key_values`, then (a) the reference's eager block on the narrowed views with the additive mask its
    `_prepare_decoder_attention_mask` builds (causal + padding + `tree_mask == 0 -> min`, here from lantern_drafter_attention_mask,
    the same recipe) and (b) one `ops.tree_attention` call on the slab views.  Batch 2 = cond / left-padded uncond row."""
    import types
    from lantern_amd.drafters.kv_cache import initialize_past_key_values
    B, Hq, d, S_max, prev = 2, 4, 128, 512, 137
    tb = generate_tree_buffers(mc_sim_7b_63, device="cuda")
    tm = tb["tree_attn_mask"]                                  # [1,1,N,N]
    N = tm.shape[-1]
    lin = types.SimpleNamespace(weight=torch.zeros(1, device="cuda"))
    fake = types.SimpleNamespace(config=types.SimpleNamespace(num_hidden_layers=2, num_key_value_heads=Hq, max_position_embeddings=S_max,
                                                              hidden_size=Hq * d, num_attention_heads=Hq),
                                 model=types.SimpleNamespace(layers=[types.SimpleNamespace(self_attn=types.SimpleNamespace(q_proj=lin))]),
                                 dtype=torch.bfloat16)
    pkv, slabs, cur = initialize_past_key_values(fake, batch_size=B)
    g = torch.Generator(device="cuda").manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g, device="cuda").to(torch.bfloat16)      # noqa: E731
    kcache, vcache = pkv[1]                                                               # second layer: a non-zero slab offset
    kcache.cat(rnd(B, Hq, prev, d)); vcache.cat(rnd(B, Hq, prev, d))                     # prefill
    q = rnd(B, Hq, N, d)                                                                  # query_states after .transpose(1, 2)
    k_all = kcache.cat(rnd(B, Hq, N, d)); v_all = vcache.cat(rnd(B, Hq, N, d))           # tree step appended, narrowed views back
    assert k_all.shape[2] == prev + N and int(cur[2]) == prev + N
    pad = 29                                                                              # uncond row: 29 left-padded positions
    attn = torch.ones(B, prev + N, dtype=torch.bool, device="cuda")
    attn[1, :pad] = False
    mask = ops.drafter_attention_mask(attn, tm, B, N, prev)                               # [B,1,N,prev+N] additive f32
    w = torch.matmul(q, k_all.transpose(2, 3)) / math.sqrt(d) + mask
    p = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    eager = torch.matmul(p, v_all).transpose(1, 2).reshape(B, N, Hq * d)
    out = ops.tree_attention(q.transpose(1, 2), kcache.data, vcache.data, ops.tree_mask_bits(tm),
                             kv_len=torch.tensor([prev + N, prev + N], device="cuda"), kv_start=torch.tensor([0, pad], device="cuda"),
                             max_kv_len=prev + N)
    err = (out.float() - eager.float()).abs()
    assert (err <= ATOL + RTOL * eager.float().abs()).all(), float(err.max())


# ---------------------------------------------------------------------------------------------------------------------------
# Vectors recorded from the REFERENCE's own eager attention (tests/golden/make_golden_attention.py: ChameleonAttention.forward of
# models/kv_variants/modeling_lumina_mgpt_kv.py run on the CPU in bf16 against the mask of _prepare_decoder_attention_mask and the
# reference's KVCache): rotated queries, caches, mask -> the attention core's output in front of o_proj.
import os

import numpy as np

_ATT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "attention.npz")
_ATT_CASES = ["static26_d128", "static58_d64_gqa", "dynamic59_d128_gqa", "dynamic59_d64"]


def _bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.bfloat16)


@pytest.mark.parametrize("name", _ATT_CASES)
def test_reference_recorded_attention(name):
    z = np.load(_ATT)
    heads, kvh, d, N, L0, s0, s1 = (int(v) for v in z[name + ".meta"])
    q = _bf16(z[name + ".q"]).cuda()                     # [B, Hq, N, d]
    k, v = _bf16(z[name + ".k"]).cuda(), _bf16(z[name + ".v"]).cuda()          # [B, Hkv, L0 + N, d]
    want = _bf16(z[name + ".out"]).float().cuda()        # [B, N, Hq * d]
    vis = torch.from_numpy(z[name + ".mask"])            # [B, 1, N, L0 + N] what the reference's additive mask lets through
    tm = torch.from_numpy(z[name + ".tree_mask"])
    B, S = q.shape[0], L0 + N
    starts = [s0, s1]
    # the mask the kernel derives from (kv_start, kv_len, tree bits) is the reference's mask, cell for cell
    for b in range(B):
        implied = torch.zeros(N, S, dtype=torch.bool)
        implied[:, starts[b]:L0] = True
        implied[:, L0:] = tm != 0
        assert torch.equal(implied, vis[b, 0]), name
    # the caches as the product holds them: S_max rows, junk behind the end
    S_max = S + 13
    kc = torch.full((B, kvh, S_max, d), 1e30, dtype=torch.bfloat16, device="cuda")
    vc = torch.full((B, kvh, S_max, d), float("nan"), dtype=torch.bfloat16, device="cuda")
    kc[:, :, :S], vc[:, :, :S] = k, v
    bits = ops.tree_mask_bits(tm.cuda())
    lens = torch.full((B,), S, dtype=torch.int64, device="cuda")
    got = ops.tree_attention(q.transpose(1, 2), kc, vc, bits, kv_len=lens, kv_start=torch.tensor(starts, dtype=torch.int64, device="cuda")).float()
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    err = (got - want).abs()
    assert (err <= ATOL + RTOL * want.abs()).all(), (name, float(err.max()))
    # ... and as close to the exact (f64) result as the reference's own bf16 pipeline
    exact = exact_attention(q.transpose(1, 2).contiguous(), kc, vc, tm.cuda(), lens, torch.tensor(starts, device="cuda"), d ** -0.5).float()
    assert float((got - exact).abs().max()) <= float((want - exact).abs().max()) + 2e-2
