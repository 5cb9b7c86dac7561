"""GPU (-m gpu): guard bands around every device buffer the op wrappers allocate.  There is no GPU address sanitizer on this pool,
so a selection of the parity tests is re-run with `torch.empty / zeros / full` (as used by lantern_amd.ops for outputs, workspaces
and staging buffers) replaced by allocations padded with 4 KiB of 0xA5 on both sides; after the kernels have run every pad byte
must still be 0xA5.  A kernel that writes before or past an output it was handed trips this even when the stray bytes land in
memory the caching allocator owns (where nothing would fault)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PAD = 4096


class GuardedAllocations:
    def __init__(self):
        self.bufs = []

    def _alloc(self, orig, fill, size, kw):
        device = kw.get("device", None)
        if device is None or "cuda" not in str(device):
            return None
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(int(s) for s in size)
        dtype = kw.get("dtype", None) or torch.get_default_dtype()
        es = torch.empty((), dtype=dtype).element_size()
        nbytes = int(np.prod(shape, dtype=np.int64)) * es if len(shape) else es
        span = (nbytes + 15) // 16 * 16
        raw = self.o_full((PAD + span + PAD,), 0xA5, dtype=torch.uint8, device=device)
        mid = raw[PAD:PAD + nbytes]
        if fill is not None:
            mid.fill_(0)
        out = mid.view(dtype).view(shape) if nbytes else self.o_empty(shape, dtype=dtype, device=device)
        if fill not in (None, 0):
            out.fill_(fill)
        self.bufs.append((raw, nbytes))
        return out

    def __enter__(self):
        self.o_empty, self.o_zeros, self.o_full = torch.empty, torch.zeros, torch.full

        def empty(*size, **kw):
            r = self._alloc(self.o_empty, None, size, kw)
            return r if r is not None else self.o_empty(*size, **kw)

        def zeros(*size, **kw):
            r = self._alloc(self.o_zeros, 0, size, kw)
            return r if r is not None else self.o_zeros(*size, **kw)

        def full(size, fill_value, **kw):
            r = self._alloc(self.o_full, fill_value, (size,), kw)
            return r if r is not None else self.o_full(size, fill_value, **kw)

        torch.empty, torch.zeros, torch.full = empty, zeros, full
        return self

    def __exit__(self, *exc):
        torch.empty, torch.zeros, torch.full = self.o_empty, self.o_zeros, self.o_full
        return False

    def check(self):
        torch.cuda.synchronize()
        assert self.bufs, "nothing was allocated under the guard"
        for raw, nbytes in self.bufs:
            span = raw.numel() - 2 * PAD
            assert bool((raw[:PAD] == 0xA5).all()), "write in front of a buffer"
            assert bool((raw[PAD + nbytes:] == 0xA5).all()), f"write past a buffer of {nbytes} bytes (span {span})"
        return len(self.bufs)


def _run(fn, *args):
    with GuardedAllocations() as g:
        fn(*args)
        n = g.check()
    assert n > 0


def test_guard_detects_a_stray_write():
    with GuardedAllocations() as g:
        t = torch.empty((10,), dtype=torch.float32, device="cuda")
        base = t.untyped_storage()
        whole = torch.tensor([], dtype=torch.uint8, device="cuda").set_(base)
        whole[PAD + 40] = 7                      # one byte past the 40-byte tensor
        with pytest.raises(AssertionError, match="past a buffer"):
            g.check()


def test_o7_window_and_dense_outputs():
    import test_gpu_window as W
    _run(W.test_cfg_window_golden, "bf16")
    _run(W.test_cfg_window_probs_and_temperature, torch.bfloat16, 4096)
    _run(W.test_cfg_window_top_p, 0.5, True)
    import test_gpu_parity as P
    _run(P.test_cfg_mask_topk_full_size_vs_oracle)


def test_evaluate_posterior_outputs():
    import test_gpu_window as W
    import test_gpu_configs as C
    import test_gpu_fuzz as F
    _run(W.test_window_full_size_lumina_pipeline_vs_oracle)
    _run(W.test_window_maximum_tree_shape, "anole")
    _run(W.test_window_more_candidates_than_prefetch_slots, "lumina", False)
    _run(C.test_c2_llamagen_dynamic_standard_verify)
    _run(C.test_c4_anole_static_lantern_pp, 5.0, 10)
    _run(C.test_window_large_k_reads_ids_from_hbm, 3000, 0.3)
    _run(F.test_static_batches_vs_oracle, "lumina", "mc_sim_7b_63", True, 100, 0.1, 1.0, 1)
    _run(F.test_static_batches_vs_oracle, "llamagen", "naive_extend_57", True, 1000, 0.05, 1.5, 10)


def test_gather_update_and_tree_outputs():
    import test_gpu_parity as P
    _run(P.test_kv_and_accept_gather_golden)
    _run(P.test_update_inference_inputs_fused_equals_separate_ops)
    _run(P.test_kv_gather_deep_paths, 16, 128)
    _run(P.test_gather_candidates_and_sample_static_golden)
    _run(P.test_bonus_token_inverse_cdf_vs_oracle)
    dyn = [i for i, sp in enumerate(P.SPECS) if sp["kind"] in ("dynamic", "greedy")][::3]
    for i in dyn[:3]:
        _run(P.test_dynamic_tree_golden, i)


def test_drafter_table_and_attention_outputs():
    import test_gpu_more as M
    import test_gpu_drafter as D
    import test_gpu_tree_attention as T
    _run(M.test_vq_table_builder)
    _run(M.test_drafter_fc_mfma_vs_oracle, 20, 4096, 1.0, True, False)
    _run(M.test_drafter_fc_mfma_vs_oracle, 20, 4096, 2.0, True, True)          # stream-K on the packed weight: workspace partials under guard bands too
    _run(D.test_linear_rows_matches_torch, 20, 4096, 4, 8192)
    _run(D.test_drafter_head_window_equals_full_head)
    _run(T.test_tree_sizes, 59)
    _run(T.test_tree_sizes, 26)
    _run(T.test_split_keys_path_matches_single_pass, None)
    _run(T.test_head_shapes_and_gqa, 64, 8, 2)
    _run(T.test_left_padding, 40)


def test_harness_step_loop_buffers():
    """The benchmark harness (pools, per-step state, logs, KV slabs -- its own allocations, handed to the C-ABI as raw pointers)."""
    import test_gpu_loop as L
    _run(L.test_harness_loop_matches_oracle_loop, "window", False, 1, 6, 36)
    _run(L.test_harness_loop_matches_oracle_loop, "window", False, 3, 6, 36)
    _run(L.test_harness_loop_matches_oracle_loop, "dense", False, 1, 6, 36)
    _run(L.test_kv_rows_follow_the_accepted_path)
