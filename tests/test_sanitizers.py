"""CPU-only sanitizer leg: the oracle's C restatement and the host-compiled field / curve templates of the bucket kernels under
AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on this pool: the CPU builds of the same code are
what can be checked).  A finding aborts the subprocess and fails the test."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(shutil.which("gcc") is None or _asan_runtime() is None, reason="gcc / libasan not available")
def test_oracle_golden_checks_under_asan_ubsan():
    """oracle/oracle.c built with -fsanitize=address,undefined (oracle/Makefile: liboracle_san.so), loaded into a fresh interpreter
    with the ASan runtime preloaded, runs the whole golden suite of the oracle (tests/test_oracle_golden.py: every reference-WASM
    vector, raw bytes) and the host-side C-ABI checks."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "liboracle_san.so"], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_san.so"), LD_PRELOAD=_asan_runtime(),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "passed" in r.stdout and "runtime error" not in (r.stdout + r.stderr)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_field_u_templates_under_asan_ubsan(tmp_path):
    """csrc/field_u.hpp + curve_u.hpp (the bucket kernels' arithmetic) compiled for the host with the sanitizers: shifts, limb
    indexing and the bound-tracked lazy arithmetic run clean on the same vectors as tests/test_field_u_host.py."""
    exe = tmp_path / "test_field_u_san"
    subprocess.run(["hipcc", "--cuda-host-only", "-O0", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I", os.path.join(ROOT, "zprize-wasm-msm_amd", "csrc"), os.path.join(ROOT, "tests", "host", "test_field_u.hip"), "-o", str(exe)],
                   check=True, timeout=900)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert r.stdout.count(": ok") == 4 and "runtime error" not in r.stderr
