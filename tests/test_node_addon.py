"""The Node.js host side (N-API addon + index.js shim) over the same C ABI."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE_DIR = os.path.join(ROOT, "zprize-wasm-msm_amd", "node")

needs_node = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists("/usr/include/node/node_api.h"),
                                reason="node / node_api.h not present on this box")


@pytest.fixture(scope="module")
def addon():
    import __graft_entry__ as ge
    ge.build_engine()
    path = ge.build_node_addon()
    assert path and os.path.exists(path)
    return path


@needs_node
def test_addon_loads_and_host_helpers_match_golden(addon):
    """No GPU needed: the addon resolves every C-ABI symbol it binds and the host-only helpers agree with golden data."""
    r = subprocess.run(["node", "test_golden.js", "--host-only"], cwd=NODE_DIR, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checks passed" in r.stdout


@needs_node
def test_addon_surface_names(addon):
    js = ("const m=require('./index.js');"
          "const e=m.instantiate.length; const names=Object.keys(m.native).sort().join(',');console.log(names);")
    r = subprocess.run(["node", "-e", js], cwd=NODE_DIR, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    for name in ("create", "multiexp", "multiexpChunk", "multiexpAsync", "multiexpChunkAsync", "normalize", "toAffine", "eq", "combine"):
        assert name in r.stdout


@needs_node
@pytest.mark.gpu
def test_addon_golden_on_gpu(addon):
    """Both JS surfaces (WASM-export-shaped over a linear memory; async buffers) reproduce every golden MSM record."""
    r = subprocess.run(["node", "test_golden.js"], cwd=NODE_DIR, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checks passed" in r.stdout
