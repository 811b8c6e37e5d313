import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Whatever test runs first finds the in-tree library: compile it if it has never been built
    (hipcc cross-compiles gfx950 without a GPU; on the GPU box the prebuilt .so travels with the tree)."""
    from svtyper_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        hip.build()


@pytest.fixture(scope="session")
def fixture_library():
    """The reference fixture's single library (tests/data/NA12878.bam.json)."""
    import json
    from svtyper_amd.evidence import LibraryTable
    with open(os.path.join(ROOT, "tests", "data", "NA12878.bam.json")) as f:
        info = json.load(f)
    lib = info["NA12878"]["libraryArray"][0]
    hist = {int(k): int(v) for k, v in lib["histogram"].items()}
    return LibraryTable.from_counter(hist, float(lib["mean"]), float(lib["sd"]), lib["library_name"])


@pytest.fixture(scope="session")
def hip_device(request):
    """Loads the HIP library and requires a device.  Without one the gpu-marked tests FAIL -- a missing device must
    never look like a green run (`pytest tests -m "not gpu"` is the selection for a box without an MI355X).  Only
    SVT_ALLOW_NO_GPU=1, set on purpose, turns the failure into a skip for a plain `pytest tests` on such a box."""
    from svtyper_amd import hip
    hip.load()
    if hip.device_count() <= 0:
        asked = request.config.getoption("-m") or ""
        if os.environ.get("SVT_ALLOW_NO_GPU") == "1" and not ("gpu" in asked and "not gpu" not in asked):
            pytest.skip("no MI355X visible (SVT_ALLOW_NO_GPU=1)")
        pytest.fail("no MI355X visible: the gpu-marked tests need the real device (select -m \"not gpu\" on a box without one)")
    return 0
