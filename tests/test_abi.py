"""The C-ABI library loads and exports every symbol include/svtyper_hip.h declares (no compute:
runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="svtyper_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svt_[a-z_]+)\s*\(", src)))


def test_header_symbols_exported():
    from svtyper_amd import hip
    hip.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), "missing export %s" % s
    assert sorted(hip.EXPORTS) == syms
    reads = declared_symbols("svtyper_reads.h")
    assert len(reads) >= 9
    for s in reads:
        assert hasattr(lib, s), "missing export %s" % s


def test_version_and_loud_failure_without_gpu():
    from svtyper_amd import hip, synth
    L = hip.load()
    assert L.svt_version() == hip.ABI_VERSION
    if hip.device_count() == 0:
        import pytest
        b = synth.make_units(10, 1, [synth.normal_library(n=20000)])
        with pytest.raises(hip.SvtyperHipError) as e:
            hip.genotype_batch(b)
        assert "no CPU fallback" in str(e.value) or "no HIP device" in str(e.value)
