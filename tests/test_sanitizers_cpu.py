"""Sanitizer leg (CPU only -- never on the GPU box): the C restatement of the oracle and the host-only work-list
planner are rebuilt under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`, `make -C csrc plan_asan`) and the tests that
drive them -- the oracle against every golden fixture, the schedule invariants, the tiler -- run once more in a child python with libasan
preloaded.  Any report (heap overflow in a top-k buffer, signed overflow in the schedule arithmetic, misaligned vector access) fails the run:
-fno-sanitize-recover makes UBSan fatal, ASan aborts by itself."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "open-hummingbird-eval_amd", "csrc")


def _libasan():
    if not shutil.which("gcc"):
        return None
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return os.path.realpath(p) if p and os.path.sep in p and os.path.exists(p) else None


def _run_under_asan(pytest_args, extra_env):
    asan = _libasan()
    if asan is None:
        pytest.skip("gcc / libasan not available")
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="4")
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + pytest_args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    return out


def test_oracle_c_restatement_under_asan_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    lib = os.path.join(ROOT, "oracle", "libhbird_oracle_asan.so")
    # (the two tests that import torch stay outside: a preloaded libasan under torch's own allocator proves nothing about our code)
    out = _run_under_asan(["tests/test_oracle_golden.py", "-k", "not g5_sample_features and not reproducible_from_the_reference and not torch_float64"], {"HBIRD_ORACLE_LIB": lib})
    assert " passed" in out


def test_work_list_planner_under_asan_ubsan():
    subprocess.run(["make", "-C", CSRC, "plan_asan"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    lib = os.path.join(ROOT, "open-hummingbird-eval_amd", "lib", "build", "libhbird_plan_asan.so")
    out = _run_under_asan(["tests/test_schedule_cpu.py", "-k", "not multi_index_row_planning"],
                          {"HBIRD_HIP_LIB": lib, "HBIRD_PLAN_ONLY": "1"})
    assert " passed" in out
    # ... and the decisions taken from the workgroups' time stamps (csrc/hbird_calibrate.cpp, in the same host-only library)
    out = _run_under_asan(["tests/test_calibrate_cpu.py"], {"HBIRD_HIP_LIB": lib, "HBIRD_PLAN_ONLY": "1"})
    assert " passed" in out
