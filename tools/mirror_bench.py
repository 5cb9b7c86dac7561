"""The drop-in API on the clock: EaLumina_mGPT.generate (what entrypoints/generate_images.py:240 calls through the solver) driven by stand-in
target / drafter models whose forwards cost (almost) nothing -- pre-generated logits, hidden states and drafter samples looked up from pools -- at the
full Chameleon vocabulary (V = 65536), the reference's default tree mc_sim_7b_63 and the 7B KV geometry (32 layers x 32 heads x 128, sequential CFG:
two caches): what remains is the mirror's own verify step -- generate_candidates, tree_decoding's post-process, evaluate_posterior, the KV /
hidden / token commit, the one host read per step -- i.e. the host + kernel cost of the reference's loop body (models/ea_model_lumina_mgpt.py:936-1015)
as this package runs it at the reference's batch of one.

usage: mirror_bench.py [steps] [kv_layers]   ->  one JSON line (us per verify step, accepted tokens per step)"""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from lantern_amd import harness as HN, ops
from lantern_amd.drafters.choices import mc_sim_7b_63
from lantern_amd.ea_model_lumina_mgpt import EaLumina_mGPT

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
L_KV = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev, bf = torch.device("cuda"), torch.bfloat16
V, H, S = HN.V, HN.HIDDEN, 16
tb = ops.tree_static_build(mc_sim_7b_63)
N = len(tb["tree_indices"])
R = int(((tb["tree_indices"][1:] - 1) // 10).max()) + 1
# pools by the bench's recipe (harness.py): CFG-consistent cond / uncond logits, drafter rows = target rows + noise, 10 samples per row
wl = HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=1, pool_steps=S, with_kv=False, max_steps=8, ep_kernel="chain"), dev)
T_PRE = 32
cond = torch.zeros((S, T_PRE, V), dtype=bf, device=dev)
unc = torch.zeros((S, T_PRE, V), dtype=bf, device=dev)
cond[:, :N], unc[:, :N] = wl.cond[:, 0], wl.uncond[:, 0]
hid = wl.hidden[:, 0]                                             # [S, 2, N, H]
hid_pad = torch.zeros((S, 2, T_PRE, H), dtype=bf, device=dev)
hid_pad[:, :, :N] = hid
orig_dense = torch.zeros((S, R, V), dtype=torch.float32, device=dev)
orig_dense[..., HN.IMG_LO:HN.IMG_HI] = wl.orig_prob[:, 0]
ss_token, ss_prob = wl.ss_token[:, 0], wl.ss_prob[:, 0]
levels = ops.tree_drafter_build(mc_sim_7b_63, 10)["tree_indices"]
n_lvl = [1] + [len(t) for t in levels]
assert sum(n_lvl) == R, (n_lvl, R)
orig_lists = [list(torch.split(orig_dense[s], n_lvl)) for s in range(S)]


class Clock:
    i = 0


class FakeHead:
    def __init__(self):
        self.weight = torch.zeros(V, H, device=dev, dtype=bf)
        self.calls = 0

    def __call__(self, hidden):          # cond pass, then uncond pass of the same step
        pool = cond if self.calls % 2 == 0 else unc
        self.calls += 1
        return pool[Clock.i % S, :hidden.shape[1]][None]


class FakeInner:
    def __init__(self):
        lin = types.SimpleNamespace(weight=torch.zeros(1, device=dev))
        self.layers = [types.SimpleNamespace(self_attn=types.SimpleNamespace(q_proj=lin)) for _ in range(L_KV)]
        self.tree_mask, self.tree_mode, self.calls = None, None, 0

    def __call__(self, input_ids=None, attention_mask=None, past_key_values=None, position_ids=None):
        j = self.calls % 2
        self.calls += 1
        return (hid_pad[Clock.i % S, j, :input_ids.shape[1]][None],)          # the KV rows of a real forward are its own cost, not the path's


class FakeDrafter:
    cfg_scale = 3.0

    def reset_kv(self):
        pass

    def init_tree(self, tree=None):
        pass

    def topK_generate(self, hidden_states, uncond_hidden_states, input_ids, attention_mask, head, logits_processors, tree_type="static"):
        s = Clock.i % S
        return ss_token[s], ss_prob[s], orig_lists[s]


head = FakeHead()
cfg = types.SimpleNamespace(num_hidden_layers=L_KV, num_key_value_heads=32, max_position_embeddings=4096, hidden_size=H, num_attention_heads=32)
base = types.SimpleNamespace(model=FakeInner(), lm_head=head, config=cfg, dtype=bf)
mdl = EaLumina_mGPT(base, FakeDrafter(), wl.table_full, cfg_mode="sequential", eagle_version=1)
orig_step = mdl._verify_step


def counted(*a, **k):
    r = orig_step(*a, **k)
    Clock.i += 1
    return r


mdl._verify_step = counted
prompt = torch.randint(9000, 12000, (1, 10), device=dev)
kw = dict(cfg_scale=3.0, top_k=2000, lantern=True, lantern_k=1000, lantern_delta=0.1, tree_choices=mc_sim_7b_63)
mdl.generate(prompt, max_new_tokens=60, **kw)          # warm-up: code objects, tree buffers, KV slabs
torch.cuda.synchronize()
Clock.i = 0
prof = None
if os.environ.get("MIRROR_PROFILE") == "1":          # (tools/mirror_profile.py: cProfile of the timed call only)
    import cProfile
    prof = cProfile.Profile()
    prof.enable()
t0 = time.perf_counter()
ids, acc = mdl.generate(prompt, max_new_tokens=steps * 3, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if prof is not None:
    import io
    import pstats
    prof.disable()
    for key in ("tottime", "cumulative"):
        s_ = io.StringIO()
        pstats.Stats(prof, stream=s_).strip_dirs().sort_stats(key).print_stats(30)
        print(s_.getvalue()[:7000], file=sys.stderr)
n = len(acc)
print(json.dumps({"workload": f"EaLumina_mGPT.generate, static tree mc_sim_7b_63, V={V}, k=1000, delta=0.1, sequential CFG, {2 * L_KV * 2} KV slabs of the 7B geometry, "
                              "stand-in target / drafter forwards (pool lookups)", "verify_steps": n, "us_per_verify_step": 1e6 * dt / max(n, 1),
                  "accepted_tokens_per_step": float(np.mean(acc)), "ep_form": mdl.ep_form}))
