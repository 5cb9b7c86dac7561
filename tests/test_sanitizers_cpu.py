"""CPU: the library's own HOST code and the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY section 5 "sanitizers"; VERDICT
round 5, item 8).  There is no GPU AddressSanitizer on the MI355X pool (tests/test_gpu_canary.py re-runs the GPU parity tests between guard
bands instead); what CAN be sanitized is everything in front of a launch:

  liblantern_hip_asan.so  (`make -C lantern_amd/csrc asan`): every .hip / .cpp source compiled --offload-host-only with -fsanitize=address,undefined
      -- the static / drafter tree builders and node tables (tree_static.cpp), lantern_verify_step's sequencing and validation, the argument
      checks and launch-shape arithmetic of every entry point.  tests/test_cabi_cpu.py and the host half of tests/test_gpu_nodes.py run against it.
  liblantern_oracle_asan.so (`make -C oracle asan`): the C restatement, through the reference-generated vectors of tests/test_oracle_golden.py.

Each runs in a child interpreter with the matching sanitizer runtime preloaded (clang's for the hipcc build, gcc's for the oracle); any report
ends the child with a non-zero status (halt_on_error, -fno-sanitize-recover)."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _run(env_extra, args, timeout=900):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    return r.stdout


@pytest.mark.timeout(1200)
@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_library_host_side_under_asan_ubsan():
    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rts:
        pytest.skip("clang's shared ASan runtime is not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "lantern_amd", "csrc"), "-s", "-j8", "asan"])
    so = os.path.join(ROOT, "lantern_amd", "liblantern_hip_asan.so")
    nm = subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout
    assert "__asan_init" in nm and "__ubsan_handle" in nm                # really instrumented
    out = _run({"LD_PRELOAD": rts[0], "LANTERN_HIP_LIBRARY": so},
               ["tests/test_cabi_cpu.py", "tests/test_gpu_nodes.py", "tests/test_loaders_cpu.py"])
    assert " passed" in out


@pytest.mark.timeout(1200)
@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not installed")
def test_oracle_under_asan_ubsan():
    rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("gcc's shared ASan runtime is not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    so = os.path.join(ROOT, "oracle", "liblantern_oracle_asan.so")
    out = _run({"LD_PRELOAD": rt, "LANTERN_ORACLE_LIBRARY": so}, ["tests/test_oracle_golden.py", "tests/test_oracle_loop_mt.py"])
    assert " passed" in out
