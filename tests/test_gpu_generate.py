"""GPU (-m gpu): `EaLumina_mGPT.generate()` / `eagenerate()` of the host mirror driven end to end by a scripted target
model and drafter (no checkpoints exist offline).  The fake target writes (token, position) into its K/V rows, so after
generation the KV cache must spell exactly the accepted token sequence at consecutive positions -- that holds only if
candidate assembly, tree positions, evaluate_posterior, the KV gather, the length bookkeeping and the uniform FIFO all
line up across steps.  Both tree types (static EAGLE-1 / dynamic EAGLE-2) and both CFG modes."""
import random
import types

import pytest
import torch

from lantern_amd import ops
from lantern_amd.drafters.choices import mc_sim_7b_63
from lantern_amd.ea_model_lumina_mgpt import EaLumina_mGPT

pytestmark = pytest.mark.gpu
V, H, HKV, DH, SMAX, M = 16384, 64, 2, 32, 512, 97


class FakeHead:
    """lm_head: logits row looked up from (token, position) carried in the hidden state -- exact on every device."""

    def __init__(self, dev):
        g = torch.Generator(device="cpu").manual_seed(5)
        self.weight = torch.zeros(V, H, device=dev, dtype=torch.bfloat16)
        t = (3.0 * torch.randn(M, V, generator=g)).to(torch.bfloat16)
        self.table = t.to(dev)

    def __call__(self, hidden):
        tok, pos = hidden[..., 0].float().long(), hidden[..., 1].float().long()
        return self.table[(tok * 7 + pos * 13) % M]


class FakeInner:
    def __init__(self, dev, n_layers=2):
        lin = types.SimpleNamespace(weight=torch.zeros(1, device=dev))
        self.layers = [types.SimpleNamespace(self_attn=types.SimpleNamespace(q_proj=lin)) for _ in range(n_layers)]
        self.tree_mask, self.tree_mode, self.dev = None, None, dev

    def __call__(self, input_ids=None, attention_mask=None, past_key_values=None, position_ids=None):
        B, T = input_ids.shape
        cur = int(past_key_values[0][0].current_length)
        if position_ids is None:
            position_ids = torch.arange(cur, cur + T, device=self.dev)[None].expand(B, T)
        position_ids = position_ids.reshape(-1, T).expand(B, T)
        hidden = torch.zeros(B, T, H, device=self.dev, dtype=torch.bfloat16)
        # token ids up to 16383 are not exact in bf16: split into two exactly representable digits
        hidden[..., 0] = (input_ids % 128).to(torch.bfloat16)
        hidden[..., 2] = (input_ids // 128).to(torch.bfloat16)
        hidden[..., 1] = (position_ids % 128).to(torch.bfloat16)
        hidden[..., 3] = (position_ids // 128).to(torch.bfloat16)
        kv = hidden[:, None, :, :DH].expand(B, HKV, T, DH).contiguous()
        for layer in past_key_values:
            for c in layer:
                c.cat(kv, dim=2)
        return (hidden,)


def decode(h):          # hidden/KV row -> (token, position)
    return (h[..., 0].float() + 128 * h[..., 2].float()).long(), (h[..., 1].float() + 128 * h[..., 3].float()).long()


class Head2(FakeHead):
    def __call__(self, hidden):
        tok, pos = decode(hidden)
        return self.table[(tok * 7 + pos * 13) % M]


class FakeDrafter:
    """EAGLE drafter stand-in with the reference's interface (init_tree / reset_kv / topK_generate).  Its tree logic runs
    on the HIP ops (O2/O3/O4/O5); its 'network' is a table lookup a little off the target's."""

    def __init__(self, dev, head):
        self.dev, self.head, self.cfg_scale = dev, head, 3.0
        g = torch.Generator(device="cpu").manual_seed(9)
        self.noise = (1.5 * torch.randn(M, V, generator=g)).to(dev)
        self.total_tokens, self.depth, self.top_k = 58, 4, 10

    def reset_kv(self):
        pass

    def init_tree(self, tree=None):
        self.tree_buffer = ops.tree_drafter_build(tree) if tree is not None else None

    def _rows(self, toks, pos):
        key = (toks * 7 + pos * 13) % M
        lg = self.head.table[key].float() + self.noise[(key + 1) % M]
        lg[..., :4] = float("-inf")
        lg[..., 8196:] = float("-inf")
        return lg

    def topK_generate(self, hidden_states, uncond_hidden_states, input_ids, attention_mask, head, logits_processors, tree_type="static"):
        last_tok, pos = input_ids[0, -1], torch.tensor(input_ids.shape[1], device=self.dev)
        if tree_type == "dynamic":
            k = self.top_k
            root = self._rows(last_tok[None], pos[None])                      # [1,V]
            ti, cu, ci, scores = ops.expand_dynamic(root[None], None, k)
            scores_l, tokens_l = [cu.reshape(-1)], [ti.reshape(-1)]
            parents_l = [torch.zeros(1, dtype=torch.int64, device=self.dev)]
            cs = torch.arange(k, device=self.dev)
            cur_tok = ti.reshape(-1)
            for d in range(self.depth):
                parents_l.append(cs + 1 + k * k * max(0, d - 1) + (k if d > 0 else 0))
                rows = self._rows(cur_tok, pos + d + 1)
                ti, cu, ci, scores = ops.expand_dynamic(rows[None], scores, k)
                cs = ci[0]
                cur_tok = ti.reshape(-1)[cs]
                scores_l.append(cu.reshape(-1))
                tokens_l.append(ti.reshape(-1))
            draft, mask, tpos, ret, nl, md = ops.tree_dynamic_finalize(torch.cat(scores_l)[None], torch.cat(tokens_l)[None],
                                                                       torch.cat(parents_l)[None], last_tok.reshape(1), k,
                                                                       self.total_tokens, sort_rows=True)
            nl, md = int(nl[0]), int(md[0])
            return draft, ret[0, :nl, :md].contiguous(), mask[:, None], tpos[0]
        counts = [1] + [len(t) for t in self.tree_buffer["tree_indices"]]
        ss_token, ss_prob, ss_op = [], [], []
        for lvl, n in enumerate(counts):
            rows = self._rows(last_tok + torch.arange(n, device=self.dev) * 31 + lvl, pos + lvl)
            kth = torch.topk(rows, 200, dim=-1).values[..., -1:]
            op = torch.softmax(rows.masked_fill(rows < kth, float("-inf")), dim=-1)
            tok = torch.multinomial(op, 10, replacement=False)
            ss_token.append(tok)
            ss_prob.append(ops.sample_static(op, tok))
            ss_op.append(op)
        return torch.cat(ss_token), torch.cat(ss_prob), ss_op


def make_model(version, cfg_mode):
    dev = torch.device("cuda")
    head = Head2(dev)
    cfg = types.SimpleNamespace(num_hidden_layers=2, num_key_value_heads=HKV, max_position_embeddings=SMAX, hidden_size=HKV * DH,
                                num_attention_heads=HKV)
    base = types.SimpleNamespace(model=FakeInner(dev), lm_head=head, config=cfg, dtype=torch.bfloat16)
    g = torch.Generator(device="cpu").manual_seed(1)
    table = ops.build_vq_table(torch.randn(8192, 8, generator=g).to(dev))
    mdl = EaLumina_mGPT(base, FakeDrafter(dev, head), table, cfg_mode=cfg_mode, eagle_version=version)
    mdl.uniform_window = 256
    return mdl


@pytest.mark.parametrize("version,cfg_mode,kernel_set", [(2, "sequential", "window"), (1, "sequential", "window"), (2, "parallel", "window"),
                                                         (1, "parallel", "window"), (1, "sequential", "dense"), (2, "parallel", "dense")])
def test_generate_end_to_end(version, cfg_mode, kernel_set):
    random.seed(1234)
    torch.manual_seed(0)
    mdl = make_model(version, cfg_mode)
    mdl.kernel_set = kernel_set
    prompt = torch.randint(9000, 12000, (1, 11), device="cuda")
    out_ids, accept = mdl.eagenerate(prompt, max_new_tokens=60, cfg_scale=3.0, top_k=200, lantern=True, lantern_k=100, lantern_delta=0.1,
                                     tree_choices=mc_sim_7b_63)
    ids = out_ids[0]
    L0 = prompt.shape[1] + 3
    assert ids[:prompt.shape[1]].equal(prompt[0]) and ids[prompt.shape[1]:L0].tolist() == [8197, 8828, 8828]
    assert ids.shape[0] == L0 + sum(accept)      # every step appends its root (the previous bonus token) + the accepted tokens
    assert all(1 <= a <= 7 for a in accept) and sum(accept) >= 60
    new = ids[L0 + 1:]                            # the very first token is sampled from the unmasked prefill logits
    assert ((new >= 4) & (new < 8196) | (new == 8803) | (new == 8196)).all()
    # the target's KV rows spell the accepted sequence: row p holds (ids[p], p)
    if cfg_mode == "parallel":
        data, cl = mdl.past_key_values_data[0], mdl.current_length_data
        n_valid = int(cl[0])
        tok, pos = decode(data[0, 0, 0, :n_valid].float())
        assert tok.tolist() == ids[:n_valid].tolist() and pos.tolist() == list(range(n_valid))
    else:
        data, cl = mdl.past_key_values_data["cond"][0], mdl.current_length_data["cond"]
        n_valid = int(cl[0])
        tok, pos = decode(data[1, 0, 1, :n_valid].float())
        assert tok.tolist() == ids[:n_valid].tolist() and pos.tolist() == list(range(n_valid))
        du, clu = mdl.past_key_values_data["uncond"][0], mdl.current_length_data["uncond"]
        nu = int(clu[0])
        assert nu == n_valid - prompt.shape[1]
        tok_u, pos_u = decode(du[0, 0, 0, :nu].float())
        assert tok_u.tolist() == ids[prompt.shape[1]:n_valid].tolist() and pos_u.tolist() == list(range(nu))
    assert n_valid == ids.shape[0]


def test_window_and_dense_kernel_sets_generate_the_same_tokens():
    """Same seeds, same scripted model: the windowed kernel set (probability windows, LDS residual, packed table) and the dense
    one (the reference's full-vocabulary rows) must walk the same accept/reject decisions and emit the same ids."""
    outs = []
    for ks in ("window", "dense"):
        random.seed(77)
        torch.manual_seed(5)
        mdl = make_model(1, "sequential")
        mdl.kernel_set = ks
        # (both forms take their bonus uniforms from the same block of 4096 per torch.rand call -- EaLumina_mGPT._bonus_uniform: one torch seed, one image)
        prompt = torch.randint(9000, 12000, (1, 9), device="cuda")
        ids, accept = mdl.eagenerate(prompt, max_new_tokens=40, cfg_scale=3.0, top_k=200, lantern=True, lantern_k=100, lantern_delta=0.1,
                                     tree_choices=mc_sim_7b_63)
        outs.append((ids.cpu(), accept))
    assert outs[0][1] == outs[1][1] and torch.equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("version,cfg_mode", [(1, "sequential"), (2, "sequential"), (2, "parallel"), (1, "parallel")])
def test_generate_with_the_drafter_model_mirror(version, cfg_mode):
    """The two mirrors plugged together: EaLumina_mGPT.generate() driving drafters.cnets.Model (real input stage, attention
    mask, decoder layer, tree loops on the HIP ops) on top of the scripted target.  The drafter's proposals are arbitrary here
    (random weights), so few are accepted -- but every interface between the classes is exercised and the KV rows must still
    spell the emitted sequence."""
    from lantern_amd.drafters import cnets
    random.seed(4321)
    torch.manual_seed(1)
    dev = torch.device("cuda")
    head = Head2(dev)
    cfg = types.SimpleNamespace(num_hidden_layers=1, num_key_value_heads=HKV, max_position_embeddings=SMAX, hidden_size=H,
                                num_attention_heads=4, intermediate_size=128, vocab_size=V, pad_token_id=None)
    base_cfg = types.SimpleNamespace(num_hidden_layers=2, num_key_value_heads=HKV, max_position_embeddings=SMAX, hidden_size=HKV * DH,
                                     num_attention_heads=HKV)
    base = types.SimpleNamespace(model=FakeInner(dev), lm_head=head, config=base_cfg, dtype=torch.bfloat16)
    drafter = cnets.Model(cfg, total_tokens=59, depth=4, top_k=10, model_type="lumina_mgpt", allow_torch_layers=True).to(dev).to(torch.bfloat16)
    g = torch.Generator(device="cpu").manual_seed(1)
    table = ops.build_vq_table(torch.randn(8192, 8, generator=g).to(dev))
    mdl = EaLumina_mGPT(base, drafter, table, cfg_mode=cfg_mode, eagle_version=version)
    mdl.uniform_window = 256
    prompt = torch.randint(9000, 12000, (1, 7), device="cuda")
    out_ids, accept = mdl.eagenerate(prompt, max_new_tokens=24, cfg_scale=3.0, top_k=200, lantern=True, lantern_k=100, lantern_delta=0.1,
                                     tree_choices=mc_sim_7b_63)
    ids = out_ids[0]
    L0 = prompt.shape[1] + 3
    assert ids.shape[0] == L0 + sum(accept) and sum(accept) >= 24 and all(1 <= a <= 7 for a in accept)
    new = ids[L0 + 1:]
    assert ((new >= 4) & (new < 8196) | (new == 8803) | (new == 8196)).all()
    if cfg_mode == "parallel":
        data, n_valid = mdl.past_key_values_data[0], int(mdl.current_length_data[0])
        tok, pos = decode(data[0, 0, 0, :n_valid].float())
    else:
        data, n_valid = mdl.past_key_values_data["cond"][0], int(mdl.current_length_data["cond"][0])
        tok, pos = decode(data[1, 0, 1, :n_valid].float())
    assert n_valid == ids.shape[0] and tok.tolist() == ids.tolist() and pos.tolist() == list(range(n_valid))
    assert drafter.stable_kv is not None and drafter.stable_kv[0][0].shape[0] == 2


class FakeLlamaGenInner:
    """LlamaGen-style base: called with `cond_idx` (the 120-position conditioning block of both CFG rows) for the prefill, then with
    `input_ids`; writes (token, position) into its K/V rows like FakeInner."""

    def __init__(self, dev, n_layers=2):
        lin = types.SimpleNamespace(weight=torch.zeros(1, device=dev))
        self.layers = [types.SimpleNamespace(self_attn=types.SimpleNamespace(q_proj=lin)) for _ in range(n_layers)]
        self.tree_mask, self.tree_mode, self.dev = None, None, dev

    def __call__(self, cond_idx=None, input_ids=None, attention_mask=None, past_key_values=None, position_ids=None):
        if cond_idx is not None:
            B, T = cond_idx.shape[:2]
            input_ids = torch.zeros(B, T, dtype=torch.long, device=self.dev)
        B, T = input_ids.shape
        cur = int(past_key_values[0][0].current_length)
        if position_ids is None:
            position_ids = torch.arange(cur, cur + T, device=self.dev)[None].expand(B, T)
        position_ids = position_ids.reshape(-1, T).expand(B, T)
        hidden = torch.zeros(B, T, H, device=self.dev, dtype=torch.bfloat16)
        hidden[..., 0] = (input_ids % 128).to(torch.bfloat16)
        hidden[..., 2] = (input_ids // 128).to(torch.bfloat16)
        hidden[..., 1] = (position_ids % 128).to(torch.bfloat16)
        hidden[..., 3] = (position_ids // 128).to(torch.bfloat16)
        kv = hidden[:, None, :, :DH].expand(B, HKV, T, DH).contiguous()
        for layer in past_key_values:
            for c in layer:
                c.cat(kv, dim=2)
        return (hidden,)


@pytest.mark.parametrize("static_tree,top_p,kernel_set", [(False, 1.0, "window"), (True, 0.9, "window"), (False, 0.9, "window"),
                                                           (True, 1.0, "dense"), (False, 0.9, "dense"), (True, 0.9, "dense")])
def test_llamagen_generate_end_to_end(static_tree, top_p, kernel_set):
    """ea_model_llamagen.EaModel.generate() (V == K, processors incl. top-p, 120-token zero prefix) with drafters.cnets.Model as the
    drafter: the KV rows must spell the emitted sequence behind the conditioning block."""
    from lantern_amd.drafters import cnets
    from lantern_amd.drafters.choices import naive_extend_57
    from lantern_amd.ea_model_llamagen import EaModel
    Vl = 4096
    random.seed(7)
    torch.manual_seed(2)
    dev = torch.device("cuda")

    class HeadL(Head2):
        def __init__(self, dev):
            g = torch.Generator(device="cpu").manual_seed(5)
            self.weight = torch.zeros(Vl, H, device=dev, dtype=torch.bfloat16)
            self.table = (3.0 * torch.randn(M, Vl, generator=g)).to(torch.bfloat16).to(dev)

    head = HeadL(dev)
    base_cfg = types.SimpleNamespace(num_hidden_layers=2, num_key_value_heads=HKV, max_position_embeddings=SMAX, hidden_size=HKV * DH,
                                     num_attention_heads=HKV)
    base = types.SimpleNamespace(model=FakeLlamaGenInner(dev), lm_head=head, config=base_cfg, dtype=torch.bfloat16)
    base.encode_prompt = lambda prompt, cfg: (torch.zeros(2, 120, H, device=dev, dtype=torch.bfloat16), None)
    dcfg = types.SimpleNamespace(num_hidden_layers=1, hidden_size=H, num_attention_heads=4, intermediate_size=128, vocab_size=Vl, pad_token_id=None)
    drafter = cnets.Model(dcfg, total_tokens=59, depth=4, top_k=10, model_type="llamagen", allow_torch_layers=True).to(dev).to(torch.bfloat16)
    drafter.init_tree()
    g = torch.Generator(device="cpu").manual_seed(1)
    table = ops.build_vq_table(torch.randn(Vl, 8, generator=g).to(dev))
    mdl = EaModel(base, drafter, table)
    mdl.uniform_window = 256
    mdl.kernel_set = kernel_set
    tokens, mean_accept, seconds = mdl.generate(prompt=["a photo"], max_length=30, temperature=1.0, top_k=300, top_p=top_p, cfg=4.0,
                                                lantern=True, lantern_k=100, lantern_delta=0.2, static_tree=static_tree,
                                                tree_choices=naive_extend_57)
    assert tokens.shape == (1, 30) and 1.0 <= mean_accept <= 7.0 and seconds > 0
    assert ((tokens >= 0) & (tokens < Vl)).all()
    data, n_valid = base.past_key_values_data[0], int(base.current_length_data[0])
    tok, pos = decode(data[0, 0, 0, :n_valid].float())
    assert pos.tolist() == list(range(n_valid)) and (tok[:120] == 0).all()
    assert tok[120:120 + 30].tolist() == tokens[0].tolist()[:n_valid - 120]


# top_p < 1 is left to the LlamaGen test: with this fake head's peaked rows top-p keeps fewer than the drafter's top_k tokens, the
# drafter then proposes finfo.min (non-image) ids exactly as the reference's does (cnets_anole.py:837-845), and the table lookup on
# such an id is an index error there and LANTERN_ST_TABLE_OOB here.
@pytest.mark.parametrize("static_tree,top_p,kernel_set", [(False, 1.0, "window"), (True, 1.0, "window"), (False, 1.0, "dense"),
                                                           (True, 1.0, "dense")])
def test_anole_generate_end_to_end(static_tree, top_p, kernel_set):
    """ea_model_anole.EaModel.generate(): token-id prompt (left-padded cond row, <pad>..<bos><boi> uncond row), first token drawn
    from the image window only, cond / uncond position ids an `input_position_diff` apart, drafter called with the Anole arguments.
    The cond K/V row must spell prompt + emitted tokens at positions 0..n-1, the uncond row must sit `diff` positions behind."""
    from lantern_amd.drafters import cnets
    from lantern_amd.drafters.choices import naive_extend_57
    from lantern_amd.ea_model_anole import BOI_ID, BOS_ID, SEP_ID, EaModel
    random.seed(11)
    torch.manual_seed(4)
    dev = torch.device("cuda")
    head = Head2(dev)
    base_cfg = types.SimpleNamespace(num_hidden_layers=2, num_key_value_heads=HKV, max_position_embeddings=SMAX, hidden_size=HKV * DH,
                                     num_attention_heads=HKV)
    base = types.SimpleNamespace(model=FakeInner(dev), lm_head=head, config=base_cfg, dtype=torch.bfloat16)
    dcfg = types.SimpleNamespace(num_hidden_layers=1, hidden_size=H, num_attention_heads=4, intermediate_size=128, vocab_size=V, pad_token_id=None)
    drafter = cnets.Model(dcfg, total_tokens=59, depth=4, top_k=10, model_type="anole", allow_torch_layers=True).to(dev).to(torch.bfloat16)
    drafter.init_tree()
    g = torch.Generator(device="cpu").manual_seed(1)
    table = ops.build_vq_table(torch.randn(8192, 8, generator=g).to(dev))
    mdl = EaModel(base, drafter, table)
    mdl.uniform_window = 256
    mdl.kernel_set = kernel_set
    prompt = [[300, 9001, 77, 5000, 12]]
    tokens, mean_accept, seconds = mdl.generate(prompt=prompt, max_length=30, temperature=1.0, top_k=300, top_p=top_p, cfg=3.0,
                                                lantern=True, lantern_k=100, lantern_delta=0.2, static_tree=static_tree,
                                                tree_choices=naive_extend_57)
    L = len(prompt[0]) + 3
    assert tokens.shape == (1, 30) and 1.0 <= mean_accept <= 7.0 and seconds > 0
    assert ((tokens >= 4) & (tokens < 8196)).all()
    data, n_valid = base.past_key_values_data[0], int(base.current_length_data[0])
    tok, pos = decode(data[0, 0, 0, :n_valid].float())
    assert pos.tolist() == list(range(n_valid))
    assert tok[:L].tolist() == [BOS_ID] + prompt[0] + [SEP_ID, BOI_ID]
    assert tok[L:L + 30].tolist() == tokens[0].tolist()[:n_valid - L]
    utok, upos = decode(data[0, 1, 0, :n_valid].float())
    assert utok[:L].tolist() == [1] * (L - 2) + [BOS_ID, BOI_ID] and upos[:L].tolist() == [0] * (L - 1) + [1]
    assert utok[L:].tolist() == tok[L:].tolist() and upos[L:].tolist() == [p - (L - 2) for p in range(L, n_valid)]
    with pytest.raises(RuntimeError, match="tokenizer"):
        mdl.generate(prompt=["a cat"], max_length=4, temperature=1.0, top_k=300, top_p=1.0, cfg=3.0, lantern=False, lantern_k=10,
                     lantern_delta=0.1, static_tree=False)
    mdl.tokenizer = types.SimpleNamespace(tokenize_text=lambda s: [ord(c) + 9000 for c in s])
    t2, _, _ = mdl.generate(prompt=["a cat"], max_length=8, temperature=1.0, top_k=300, top_p=1.0, cfg=3.0, lantern=False, lantern_k=10,
                            lantern_delta=0.1, static_tree=False)
    assert t2.shape == (1, 8)
