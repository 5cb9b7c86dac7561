"""CPU: the C-ABI library loads, exports every symbol include/lantern_hip.h declares, validates
arguments, and its HOST entry points (static tree builders) match the golden vectors.
No device compute is issued here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers as H
from lantern_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TREES = [str(x) for x in H.load("trees.npz")["names"]] + [str(x) for x in H.load("trees_random.npz")["names"]]


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "lantern_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(lantern_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no prototypes found"
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in lantern_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared
    assert L.lantern_version() == 200


def test_param_struct_layout_matches_header():
    # ctypes mirror vs the C struct: field order/size (a drift here corrupts every launch)
    assert C.sizeof(_lib.EpParams) == 4 * 11 + 32 + 4 * 5 + 4 + 4 + 8 + 4 * 4
    assert C.sizeof(_lib.EpBuffers) == 20 * 8
    assert C.sizeof(_lib.EpWindow) == 8 + 8 + 8 + 5 * 8 + 8 + 4 * 8 + 6 * 4 + 2 * 8 + 8 + 2 * 8          # (+ verdict_host, + turn / turn_wait)
    assert C.sizeof(_lib.EpNodes) == 8 + 8 + 6 * 4 + 8 + 8 + 2 * 4


def test_argument_validation_without_gpu():
    L = _lib.lib()
    prm = _lib.EpParams()
    buf = _lib.EpBuffers()
    prm.B, prm.P, prm.D, prm.V = 1, 4, 3, 1022  # V not a multiple of 4
    rc = L.lantern_evaluate_posterior(C.byref(prm), C.byref(buf), None)
    assert rc == -1 and b"multiple of 4" in L.lantern_last_error()
    rc = L.lantern_cfg_mask_topk(None, None, 0, 1, 16, C.c_float(1.0), 0, None, C.c_int64(0), 1, 1, 0, 1, 0, 0, 0, None, 0, None, None)
    assert rc == -1


@pytest.mark.parametrize("name", TREES)
def test_static_tree_host_builder(name):
    g = H.tree_buffers(name)
    o = ops.tree_static_build(H.tree_choices(name))
    assert np.array_equal(o["tree_attn_mask"], g["mask"])
    assert np.array_equal(o["tree_indices"], g["tree_indices"])
    assert np.array_equal(o["tree_position_ids"], g["pos"])
    assert np.array_equal(o["retrieve_indices"], g["retrieve"])
    assert np.array_equal(o["p_indices"], g["p_indices"])
    assert np.array_equal(o["b_off"], g["b_off"])
    assert np.array_equal(o["b_idx"], g["b_idx"])


@pytest.mark.parametrize("name", TREES)
def test_drafter_tree_host_builder(name):
    g = H.tree_buffers(name)
    o = ops.tree_drafter_build(H.tree_choices(name))
    L = int(g["d_levels"][0])
    assert len(o["tree_indices"]) == L
    for l in range(L):
        assert np.array_equal(o["attn_mask"][l], g[f"d_mask{l}"])
        assert np.array_equal(o["tree_indices"][l], g[f"d_ti{l}"])
        assert o["repeat_nums"][l] == g[f"d_rep{l}"].tolist()


def test_malformed_tree_is_rejected():
    with pytest.raises(_lib.LanternError):
        ops.tree_static_build([[0], [1, 0, 0]])  # [1,0] missing


def test_device_ops_refuse_cpu_tensors():
    import torch
    with pytest.raises(_lib.LanternError):
        ops.cfg_mask_topk(torch.zeros(1, 16), torch.zeros(1, 16), 1.0)


def test_empty_batches_are_ok_and_launch_nothing():
    """B = 0 / rows = 0 / no slabs: every batched entry returns LANTERN_OK before touching the device (works without a GPU)."""
    import ctypes as C
    L = _lib.lib()
    one = C.c_void_p(8)          # any non-null pointer: nothing is dereferenced for an empty batch
    prm, buf, win = _lib.EpParams(), _lib.EpBuffers(), _lib.EpWindow()
    prm.B, prm.P, prm.D, prm.V = 0, 15, 6, 65536
    assert L.lantern_evaluate_posterior_window(C.byref(prm), C.byref(buf), C.byref(win), None) == 0
    assert L.lantern_evaluate_posterior(C.byref(prm), C.byref(buf), None) == 0
    assert L.lantern_cfg_mask_topk_window(one, one, 1, 0, 16384, C.c_float(3.0), 0, None, C.c_int64(0), 48, 48, 0, 16384, 8803, 8196, 0, None,
                                          0, 0, 16384, one, one, 0, C.c_float(1.0), C.c_float(1.0), None) == 0
    assert L.lantern_kv_gather(one, one, one, 0, 2, C.c_int64(64), C.c_int64(128), C.c_int64(128), one, 0, 15, 6, one, one, None, None) == 0
    assert L.lantern_accept_gather(None, 2, 0, 2, 26, 4096, one, 0, 15, 6, None, one, one, None, 0, None, None, None, None, None) == 0
    assert L.lantern_gather_candidates(one, None, one, one, one, 0, 110, 26, 15, 6, one, one, None, None) == 0


def test_header_is_plain_c(tmp_path):
    """include/lantern_hip.h is the FFI contract: it must compile as C99 on its own (no C++ or torch types)."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "lantern_hip.h"\nint main(void) { return sizeof(lantern_ep_params) + sizeof(lantern_ep_buffers) + sizeof(lantern_ep_window) > 0 ? 0 : 1; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "hdr.o")])


def test_uniform_fifo_replays_the_python_stream_in_order():
    """verify.UniformFifo: the device window + cursor must hand the acceptance tests exactly the sequence `random.random()`
    would have produced, across refills, whatever the (data-dependent) number of draws per step is."""
    import random
    import torch
    from lantern_amd.verify import UniformFifo
    ref = random.Random(99)
    expected = [ref.random() for _ in range(5000)]
    fifo = UniformFifo(torch.device("cpu"), window=64, rng=random.Random(99))
    rs = random.Random(5)
    consumed = []
    for _ in range(400):
        upper = rs.randint(1, 25)                 # what the host reserves: an upper bound of this step's draws
        fifo.reserve(upper)
        used = rs.randint(0, upper)               # what the kernel really consumed
        c = int(fifo.cursor[0])
        consumed += fifo.buf[0, c:c + used].tolist()
        fifo.cursor += used
    assert consumed == expected[:len(consumed)] and len(consumed) > 1000


def test_step_group_layout_matches_the_c_struct(tmp_path):
    """ctypes mirror of lantern_step_group / lantern_ep_nodes vs the C compiler's layout of include/lantern_hip.h (a drift here
    would hand every kernel of lantern_verify_step the wrong pointers)."""
    import subprocess
    wfields = ["row_hot", "rows_kind", "raw_uncond", "raw_pos_base", "raw_cfg", "raw_eos_id", "raw_probs", "raw_pre", "verdict_host", "turn", "turn_wait"]
    fields = ["stream", "B", "tree_cand", "cond", "pos_base", "w_latent", "seq_len", "temperature", "ep", "ep_buf", "ep_win", "nodes",
              "slab_ptrs", "n_slabs", "outer", "hidden", "H", "node_list", "n_list", "flags", "hidden_uncond", "ids_buf", "ids_stride", "ids_len", "prepare_next", "turn", "turn_group", "turn_groups", "turn_wait", "dyn", "greedy", "dense", "row_ready", "row_epoch"]
    src = tmp_path / "layout.c"
    gfields = [f for f, _ in _lib.StepGreedy._fields_]
    dfields = [f for f, _ in _lib.StepDense._fields_]
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lantern_hip.h"\nint main(void){printf("%zu %zu", sizeof(lantern_step_group), sizeof(lantern_ep_nodes));\n'
                   + "".join(f'printf(" %zu", offsetof(lantern_step_group, {f}));\n' for f in fields)
                   + 'printf(" %zu", sizeof(lantern_ep_window));\n' + "".join(f'printf(" %zu", offsetof(lantern_ep_window, {f}));\n' for f in wfields)
                   + 'printf(" %zu", sizeof(lantern_step_greedy));\n' + "".join(f'printf(" %zu", offsetof(lantern_step_greedy, {f}));\n' for f in gfields)
                   + 'printf(" %zu", sizeof(lantern_step_dense));\n' + "".join(f'printf(" %zu", offsetof(lantern_step_dense, {f}));\n' for f in dfields) + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == C.sizeof(_lib.StepGroup) and out[1] == C.sizeof(_lib.EpNodes)
    nf = len(fields)
    assert out[2:2 + nf] == [getattr(_lib.StepGroup, f).offset for f in fields]
    nw = len(wfields)
    assert out[2 + nf] == C.sizeof(_lib.EpWindow) and out[3 + nf:3 + nf + nw] == [getattr(_lib.EpWindow, f).offset for f in wfields]
    ng = len(gfields)
    assert out[3 + nf + nw] == C.sizeof(_lib.StepGreedy) and out[4 + nf + nw:4 + nf + nw + ng] == [getattr(_lib.StepGreedy, f).offset for f in gfields]
    assert out[4 + nf + nw + ng] == C.sizeof(_lib.StepDense) and out[5 + nf + nw + ng:] == [getattr(_lib.StepDense, f).offset for f in dfields]


def test_draft_depth_args_layout_matches_the_c_struct(tmp_path):
    """ctypes mirror of lantern_draft_depth_args vs the C compiler's layout of include/lantern_hip.h: every field's offset."""
    import subprocess
    fields = [f[0] for f in _lib.DraftDepthArgs._fields_]
    src = tmp_path / "layout_dd.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lantern_hip.h"\nint main(void){printf("%zu", sizeof(lantern_draft_depth_args));\n'
                   + "".join(f'printf(" %zu", offsetof(lantern_draft_depth_args, {f}));\n' for f in fields) + "return 0;}\n")
    exe = tmp_path / "layout_dd"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == C.sizeof(_lib.DraftDepthArgs)
    assert out[1:] == [getattr(_lib.DraftDepthArgs, f).offset for f in fields]


def test_draft_depth_validates_without_gpu():
    L = _lib.lib()
    a = _lib.DraftDepthArgs()
    assert L.lantern_draft_depth(None) < 0
    a.B, a.T, a.H, a.n_q_heads, a.n_kv_heads, a.head_dim, a.inter = 3, 10, 4096, 32, 32, 128, 11008
    assert L.lantern_draft_depth(C.byref(a)) < 0 and b"B = 2 rows" in L.lantern_last_error()
    a.B = 2
    assert L.lantern_draft_depth(C.byref(a)) < 0 and b"null weight" in L.lantern_last_error()


def test_round2_entry_points_validate_without_gpu():
    """The one-call step, the step preparation and the node form: argument errors and empty batches come back as codes + messages before
    anything touches a device (no GPU here)."""
    L = _lib.lib()
    assert L.lantern_verify_step(None, 1) == -1
    assert L.lantern_verify_step(None, 0) == -1 or L.lantern_verify_step((_lib.StepGroup * 1)(), 0) == 0
    g = _lib.StepGroup()
    g.ep.B, g.ep.P, g.ep.D, g.ep.V = 2, 15, 6, 65536
    assert L.lantern_prepare_step(C.byref(g)) == -1 and b"node list" in L.lantern_last_error()
    prm, buf, win, nodes = _lib.EpParams(), _lib.EpBuffers(), _lib.EpWindow(), _lib.EpNodes()
    prm.B, prm.P, prm.D, prm.V = 1, 15, 6, 65536
    assert L.lantern_evaluate_posterior_nodes(C.byref(prm), C.byref(buf), C.byref(win), C.byref(nodes), None) == -1      # no tables
    # a failing stage of the step names its group: a dynamic group without its score pools
    arr = (_lib.StepGroup * 2)()
    dyn = _lib.StepDynamic()
    arr[0].B, arr[0].N, arr[0].P, arr[0].D = 2, 60, 20, 7
    arr[0].dyn = C.pointer(dyn)
    assert L.lantern_verify_step(arr, 2) == -1
    msg = L.lantern_last_error()
    assert msg.startswith(b"verify_step: group 0, tree_dynamic_candidates: "), msg
    arr[0].dyn = None
    arr[0].ss_token = 0x1000          # (any non-null address: a static tree whose candidates this call assembles; nothing is dereferenced before the check fails)
    assert L.lantern_verify_step(arr, 2) == -1 and L.lantern_last_error().startswith(b"verify_step: group 0, gather_candidates: ")
    arr[0].ss_token = None            # no dynamic block, no sample list, no flag: a half-filled group is refused at O6, not evaluated on whatever `cand` holds
    assert L.lantern_verify_step(arr, 2) == -1 and L.lantern_last_error().startswith(b"verify_step: group 0, gather_candidates: ")
    arr[0].flags = _lib.STEP_CANDIDATES_READY          # "the caller assembled the candidates" said explicitly -- and then `cand` / `retrieve` must be there
    assert L.lantern_verify_step(arr, 2) == -1 and b"LANTERN_STEP_CANDIDATES_READY needs cand" in L.lantern_last_error()
    arr[0].cand = arr[0].retrieve = 0x1000             # (never dereferenced: the next stage's argument check fails first)
    assert L.lantern_verify_step(arr, 1) == -1 and L.lantern_last_error().startswith(b"verify_step: group 0, evaluate_posterior: ")
    assert L.lantern_verify_step(arr, 2) == -1 and L.lantern_last_error().startswith(b"verify_step: group 1, gather_candidates: ")      # the zero-initialised second group
    # prepare_step's three forms: a LANTERN_MODEL_PLAIN (LlamaGen) group is taken on its whole 16384-id vocabulary with dynamic trees only
    one = 0x1000                      # any non-null address: the argument check runs on the host, nothing is dereferenced or launched before it fails
    nl = (C.c_int32 * 4)(0, 1, 0, 1)
    g = _lib.StepGroup()
    g.node_list, g.n_list, g.N = C.cast(nl, C.c_void_p).value, 2, 59
    g.cond = g.uncond = g.out_win = g.row_hot = one
    g.dtype, g.model, g.V, g.win_lo, g.win_len, g.img_lo, g.img_hi = 1, 0, 16384, 0, 16384, 0, 16384
    g.out_kind, g.temperature, g.top_p = 1, 1.0, 1.0                       # LANTERN_ROWS_PROBS
    assert L.lantern_prepare_step(C.byref(g)) == -1 and b"LlamaGen dynamic trees" in L.lantern_last_error()      # static tree: refused
    g.dyn = C.pointer(dyn)
    assert L.lantern_prepare_step(C.byref(g)) == -1 and b"dynamic-tree buffers missing" in L.lantern_last_error()   # the form is accepted, the (empty) tree block is not
    g.win_len = 8192
    assert L.lantern_prepare_step(C.byref(g)) == -1 and b"LlamaGen dynamic trees" in L.lantern_last_error()


def test_dense_step_block_is_validated_before_anything_is_launched():
    """lantern_step_group.dense (the dense kernel set inside lantern_verify_step): its two buffers, token and u_bonus together, and none of the stages it
    cannot be combined with -- checked on the host, up front."""
    L = _lib.lib()
    g, q = _lib.StepGroup(), _lib.StepDense()
    g.dense = C.pointer(q)
    assert L.lantern_verify_step(C.byref(g), 1) == -1 and L.lantern_last_error().startswith(b"verify_step: group 0, dense: dense step")
    q.logits = q.sample_p = 0x1000
    q.token = 0x1000                   # (without its uniform)
    assert L.lantern_verify_step(C.byref(g), 1) == -1 and b"token and u_bonus together" in L.lantern_last_error()
    q.u_bonus = 0x1000
    nl = (C.c_int32 * 2)(0, 1)
    g.node_list, g.n_list = C.cast(nl, C.c_void_p).value, 2
    assert L.lantern_verify_step(C.byref(g), 1) == -1 and b"node_list" in L.lantern_last_error()
    g.node_list, g.n_list = None, 0
    assert L.lantern_verify_step(C.byref(g), 1) == -1 and L.lantern_last_error().startswith(b"verify_step: group 0, gather_candidates: ")   # the block itself is accepted


def test_fused_prepare_flag_is_validated_before_anything_is_launched():
    """LANTERN_STEP_FUSED_PREPARE (the prepare stage inside the chain launch): a static-tree group with a node list and the row_ready words, on the chain kernel."""
    L = _lib.lib()
    g = _lib.StepGroup()
    g.flags = _lib.STEP_FUSED_PREPARE
    assert L.lantern_verify_step(C.byref(g), 1) == -1 and L.lantern_last_error().startswith(b"verify_step: group 0, fused prepare: LANTERN_STEP_FUSED_PREPARE")
    nl = (C.c_int32 * 2)(0, 1)
    g.node_list, g.n_list, g.row_ready = C.cast(nl, C.c_void_p).value, 2, 0x1000
    g.flags = _lib.STEP_FUSED_PREPARE | _lib.STEP_PREPARED
    assert L.lantern_verify_step(C.byref(g), 1) == -1 and b"LANTERN_STEP_FUSED_PREPARE" in L.lantern_last_error()
    g.flags = _lib.STEP_FUSED_PREPARE
    q = _lib.StepDense()
    g.dense = C.pointer(q)
    assert L.lantern_verify_step(C.byref(g), 1) == -1          # (the dense block's own check, or the flag's: either way nothing is launched)
    g.dense = None
    assert L.lantern_verify_step(C.byref(g), 1) == -1 and L.lantern_last_error().startswith(b"verify_step: group 0, fused prepare: prepare_step")   # the flag is accepted, the empty group is not


def test_linear_rows_packed_validates_without_gpu():
    """lantern_linear_rows_packed (the drafter layer's GEMMs at any row count, on the packed weights): the argument checks run on the host --
    a K that is no multiple of the 64-element bricks, an unknown epilogue, a residual epilogue without its residual, a gate / up epilogue without
    the pair's row count; zero rows are a no-op."""
    L = _lib.lib()
    one = C.c_void_p(0x1000)          # any non-null address: nothing is dereferenced or launched before a check fails
    f = L.lantern_linear_rows_packed
    assert f(None, one, None, 64, 128, 32, one, 32, 0, None, 0, 0, None) == -1 and b"null buffer" in L.lantern_last_error()
    assert f(one, one, None, 64, 100, 32, one, 32, 0, None, 0, 0, None) == -1 and b"multiple of 64" in L.lantern_last_error()
    assert f(one, one, None, 64, 128, 32, one, 16, 0, None, 0, 0, None) == -1          # out rows shorter than n_rows
    assert f(one, one, None, 64, 128, 32, one, 32, 7, None, 0, 0, None) == -1 and b"epilogue 7" in L.lantern_last_error()
    assert f(one, one, None, 64, 128, 32, one, 32, 1, None, 0, 0, None) == -1 and b"residual" in L.lantern_last_error()
    assert f(one, one, None, 64, 128, 32, one, 32, 2, None, 0, 0, None) == -1 and b"pair_rows" in L.lantern_last_error()
    assert f(one, one, None, 0, 128, 32, one, 32, 0, None, 0, 0, None) == 0             # no rows: nothing to do


def test_tuning_values_are_an_explicit_api_and_the_library_reads_no_environment():
    """kernel-instance / launch-shape choices go through lantern_tuning_set (process-wide ints, defaults = the product
    path); the shared library does not import getenv at all."""
    import subprocess
    names = _lib.tuning_names()
    assert names == ["epw_tp", "epw_tp4", "epw_tp_raw", "epw_spec", "epw_occ2", "o7_nt", "prep_nt", "kv_u", "kv_ks", "kv_variant", "gemm_tiled_from",
                     "sk_groups", "sk_whole_mb", "sk_nt_min_mb", "ta_splits", "ta_min_tiles", "epw_tp_lg", "epw_fused_helpers"]
    defaults = {n: _lib.get_tuning(n) for n in names}
    assert (defaults["epw_tp"], defaults["epw_tp4"], defaults["epw_tp_raw"], defaults["epw_spec"], defaults["kv_ks"], defaults["sk_whole_mb"]) == (5, 1, 256, 2, 4, 40)
    _lib.set_tuning("epw_tp_raw", 512)
    assert _lib.get_tuning("epw_tp_raw") == 512
    assert _lib.tuning_from_env({"LANTERN_KV_KS": "2", "LANTERN_UNRELATED": "7", "LANTERN_TA_SPLITS": "x"}) == {"kv_ks": 2}
    _lib.lib().lantern_tuning_reset()
    assert {n: _lib.get_tuning(n) for n in names} == defaults
    with pytest.raises(_lib.LanternError):
        _lib.set_tuning("no_such_value", 1)
    nm = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True)
    assert nm.returncode == 0 and "getenv" not in nm.stdout
