import os
import subprocess
import sys

import pytest

# torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Whichever HIP runtime is loaded first serves the
# whole process, and two copies cannot both own the GPU: import torch BEFORE libcrt_hip.so is loaded so that the tests
# that hand torch tensors to the C ABI share one runtime with it.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is plumbing for a few tests only
    torch = None

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure(path, makedir):
    """Build a CPU-side test artefact if it is missing (the HIP library itself is built by __graft_entry__.build())."""
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", makedir], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    _ensure(os.path.join(REPO, "oracle", "liboracle_crt.so"), os.path.join(REPO, "oracle"))
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def hostsim():
    _ensure(os.path.join(REPO, "tests", "hostsim", "libhostsim.so"), os.path.join(REPO, "tests", "hostsim"))
    from tests import helpers
    return helpers.HostSim()


@pytest.fixture(scope="session")
def capi():
    from cudaraytracer_amd import capi as c
    c.lib()  # raises loudly if libcrt_hip.so is not built
    # the library reads its CRT_* tuning knobs at commit; tests switch kernel variants between renders of one scene
    c.RELOAD_TUNING_EACH_RENDER = True
    return c


@pytest.fixture(scope="session")
def scenes(capi):
    from cudaraytracer_amd import scenes as s
    return s


@pytest.fixture(params=["0", "2", "3", "4"], ids=["one-wave-groups", "lds-node-copy", "whole-tree-in-lds", "split-16-4"])
def node_cache(request, monkeypatch):
    """Runs a GPU test with every kernel variant: one wave per workgroup (CRT_NODE_CACHE=0), four waves sharing an LDS copy
    of the top of the tree (CRT_NODE_CACHE=2: always), one 16-wave workgroup per CU holding the whole node array in LDS
    (CRT_NODE_CACHE=3: small trees only; larger ones fall back to the four-wave variant), and the two-launch frame of one
    16-wave and one 4-wave workgroup per CU on one work queue (CRT_NODE_CACHE=4; trees deeper than 24 levels fall back).  The
    default, 1, chooses by the frame: test-sized frames get single waves, the benchmark's the split."""
    monkeypatch.setenv("CRT_NODE_CACHE", request.param)
    return request.param


@pytest.fixture(params=["1", "2"], ids=["slabs", "rings"])
def chunk_mode(request, monkeypatch):
    """Runs a GPU test with both ways of rendering sample chunks (crt_api.cpp: plan_chunks): CRT_CHUNK_MODE=1, every chunk's raw
    sum to a scratch of pixels x chunks and a second kernel that adds a pixel's chunk sums (what the default takes while the
    scratch fits its budget); CRT_CHUNK_MODE=2, one sample per work unit and the wave that rendered a block of pixels adds
    it up itself through its ring (what frames too large for slabs and crt_render_noalloc take).  Same frame bit for bit."""
    monkeypatch.setenv("CRT_CHUNK_MODE", request.param)
    return request.param
