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
    """Loads the HIP library and requires a device.  A run that ASKED for the gpu tests (`-m gpu`, or
    SVT_REQUIRE_GPU=1) fails loudly without one -- a missing device must never look like a green run; a plain
    `pytest tests` on a box without an MI355X skips the gpu-marked tests instead of erroring at the first one."""
    from svtyper_amd import hip
    hip.load()
    if hip.device_count() <= 0:
        asked = request.config.getoption("-m") or ""
        if os.environ.get("SVT_REQUIRE_GPU") == "1" or ("gpu" in asked and "not gpu" not in asked):
            pytest.fail("no MI355X visible: the gpu-marked tests need the real device")
        pytest.skip("no MI355X visible (run with -m gpu or SVT_REQUIRE_GPU=1 to make this a failure)")
    return 0
