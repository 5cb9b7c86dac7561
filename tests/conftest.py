import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _tuning_from_the_callers_environment():
    """Measurement scripts (tools/run/tp_try*.sh) run parts of the GPU suite under other kernel instances by setting LANTERN_<NAME>=<int> for pytest:
    the library itself reads no environment variable, so the variables become explicit lantern_tuning_set calls here (tests only)."""
    import torch
    if torch.cuda.is_available() and any(k.startswith("LANTERN_") and k[8:].lower() in
                                         ("epw_tp", "epw_tp4", "epw_tp_raw", "epw_spec", "epw_occ2", "o7_nt", "prep_nt", "kv_u", "kv_ks", "kv_variant",
                                          "gemm_tiled_from", "sk_groups", "sk_whole_mb", "sk_nt_min_mb", "ta_splits", "ta_min_tiles", "epw_tp_lg", "epw_fused_helpers") for k in os.environ):
        from lantern_amd import _lib
        _lib.tuning_from_env()
    yield
