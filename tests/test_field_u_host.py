"""Unsaturated field / curve code of the bucket kernels (csrc/field_u.hpp, csrc/curve_u.hpp), exercised on the CPU:
the same templates are compiled for the host only and compared with the packed 64-bit host arithmetic that the
oracle-pinned GPU tests rely on (mixed additions with signs, doubling / opposite / infinity cases, full additions,
long dependent chains; both curves, G1 and G2)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_field_u_against_host_field(tmp_path):
    exe = tmp_path / "test_field_u"
    subprocess.run(["hipcc", "--cuda-host-only", "-O0", "-std=c++17", "-I", os.path.join(ROOT, "zprize-wasm-msm_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "test_field_u.hip"), "-o", str(exe)], check=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(": ok") == 4, r.stdout


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_regrown_buffer_is_not_taken_for_still_zero(tmp_path):
    """csrc/engine_ctx.hpp: ensure() through an allocator that returns the same address when a buffer grows (host only)."""
    exe = tmp_path / "test_devbuf"
    subprocess.run(["hipcc", "--cuda-host-only", "-O0", "-std=c++17", "-I", os.path.join(ROOT, "zprize-wasm-msm_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "test_devbuf.hip"), "-o", str(exe)], check=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "devbuf: ok" in r.stdout
