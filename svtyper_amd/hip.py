"""ctypes binding of the gfx950 C-ABI library (include/svtyper_hip.h).

This is the ONLY compute backend of the package: if ``libsvtyper_hip.so`` is missing or no
MI355X is visible, every entry point raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from .evidence import GT_BLANK, CEvidenceBatch, CPackedEvidence, EvidenceBatch, Results

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVTYPER_HIP_LIB") or os.path.join(_HERE, "csrc", "libsvtyper_hip.so")
ABI_VERSION = 18

EXPORTS = (
    "svt_version", "svt_device_count", "svt_last_error", "svt_batch_create", "svt_batch_create_segments", "svt_batch_create_from_fragments",
    "svt_batch_genotype",
    "svt_batch_genotype_n", "svt_batch_sync", "svt_batch_genotype_timed", "svt_batch_tune_placement", "svt_batch_results", "svt_batch_device_results",
    "svt_batch_bind_device_results", "svt_batch_bind_device_results2", "svt_batch_result_order", "svt_batch_bytes", "svt_batch_layout", "svt_batch_site_qual",
    "svt_batch_stream", "svt_batch_destroy", "svt_trim", "svt_bayes_gt", "svt_genotype_counts", "svt_genotype", "svt_genotype_multi", "svt_shard_bounds",
    "svt_pinned_alloc", "svt_pinned_free", "svt_pack_evidence", "svt_packed_free", "svt_batch_create_packed", "svt_genotype_packed",
    "svt_format_results", "svt_format_free", "svt_results_host_sq", "svt_batch_result_bytes", "svt_batch_result_slots", "svt_results_expand96",
    "svt_genotype_packed_from_records", "svt_chunk_bounds",
)

_lib: Optional[C.CDLL] = None


class SvtyperHipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import glob
    srcs = [f for pat in ("*.hip", "*.cpp", "*.h") for f in glob.glob(os.path.join(_HERE, "csrc", pat))]
    srcs += glob.glob(os.path.join(_HERE, "..", "include", "*.h"))
    stale = not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-B", "libsvtyper_hip.so"])
    return LIB_PATH


def load() -> C.CDLL:
    """dlopen the library and declare prototypes.  Raises if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SvtyperHipError(
            "HIP extension not built: %s is missing (run `python -c 'import __graft_entry__ as g; "
            "g.build()'`); svtyper_amd has no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.svt_version.restype = C.c_int
    L.svt_device_count.restype = C.c_int
    L.svt_last_error.restype = C.c_char_p
    L.svt_batch_create.restype = C.c_int
    L.svt_batch_create.argtypes = [C.POINTER(CEvidenceBatch), C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
    L.svt_batch_create_segments.restype = C.c_int
    L.svt_batch_create_segments.argtypes = [C.POINTER(CEvidenceBatch), C.c_void_p, C.c_uint32, C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
    L.svt_batch_create_from_fragments.restype = C.c_int
    L.svt_batch_create_from_fragments.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.POINTER(C.c_void_p)]
    L.svt_batch_genotype.restype = C.c_int
    L.svt_batch_genotype.argtypes = [C.c_void_p, C.c_int]
    L.svt_batch_genotype_n.restype = C.c_int
    L.svt_batch_genotype_n.argtypes = [C.c_void_p, C.c_int]
    L.svt_batch_sync.restype = C.c_int
    L.svt_batch_sync.argtypes = [C.c_void_p]
    L.svt_batch_genotype_timed.restype = C.c_int
    L.svt_batch_genotype_timed.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.svt_batch_tune_placement.restype = C.c_int
    L.svt_batch_tune_placement.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.svt_batch_results.restype = C.c_int
    L.svt_batch_results.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.svt_batch_result_order.restype = C.c_int
    L.svt_batch_result_order.argtypes = [C.c_void_p, C.c_uint32]
    L.svt_batch_device_results.restype = C.c_int
    L.svt_batch_device_results.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.svt_batch_bind_device_results.restype = C.c_int
    L.svt_batch_bind_device_results.argtypes = [C.c_void_p, C.c_void_p]
    L.svt_batch_bind_device_results2.restype = C.c_int
    L.svt_batch_bind_device_results2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.svt_format_results.restype = C.c_int
    L.svt_format_results.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_void_p)]
    L.svt_format_free.restype = None
    L.svt_format_free.argtypes = [C.c_void_p, C.c_void_p]
    L.svt_batch_site_qual.restype = C.c_int
    L.svt_batch_site_qual.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64]
    L.svt_batch_result_bytes.restype = C.c_uint32
    L.svt_batch_result_bytes.argtypes = [C.c_void_p]
    L.svt_results_expand96.restype = C.c_int
    L.svt_results_expand96.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    L.svt_batch_result_slots.restype = C.c_uint64
    L.svt_batch_result_slots.argtypes = [C.c_void_p]
    L.svt_batch_layout.restype = C.c_int
    L.svt_batch_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.svt_batch_bytes.restype = C.c_int
    L.svt_batch_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.svt_batch_stream.restype = C.c_void_p
    L.svt_batch_stream.argtypes = [C.c_void_p]
    L.svt_batch_destroy.restype = None
    L.svt_batch_destroy.argtypes = [C.c_void_p]
    L.svt_trim.restype = None
    L.svt_bayes_gt.restype = C.c_int
    L.svt_bayes_gt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
    L.svt_genotype_multi.restype = C.c_int
    L.svt_genotype_multi.argtypes = [C.POINTER(CEvidenceBatch), C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_uint32, C.c_uint]
    L.svt_shard_bounds.restype = C.c_int
    L.svt_shard_bounds.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_void_p]
    L.svt_chunk_bounds.restype = C.c_int
    L.svt_chunk_bounds.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.svt_pinned_alloc.restype = C.c_void_p
    L.svt_pinned_alloc.argtypes = [C.c_size_t]
    L.svt_pinned_free.restype = None
    L.svt_pinned_free.argtypes = [C.c_void_p]
    L.svt_pack_evidence.restype = C.c_int
    L.svt_pack_evidence.argtypes = [C.POINTER(CEvidenceBatch), C.POINTER(C.POINTER(CPackedEvidence))]
    L.svt_packed_free.restype = None
    L.svt_packed_free.argtypes = [C.POINTER(CPackedEvidence)]
    L.svt_batch_create_packed.restype = C.c_int
    L.svt_batch_create_packed.argtypes = [C.POINTER(CPackedEvidence), C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
    L.svt_genotype_packed.restype = C.c_int
    L.svt_genotype_packed.argtypes = [C.POINTER(CPackedEvidence), C.c_void_p, C.c_int, C.c_uint]
    L.svt_genotype_packed_from_records.restype = C.c_int
    L.svt_genotype_packed_from_records.argtypes = [C.POINTER(CEvidenceBatch), C.c_void_p, C.c_int, C.c_uint]
    L.svt_results_host_sq.restype = C.c_int
    L.svt_results_host_sq.argtypes = [C.c_void_p, C.c_uint64]
    L.svt_genotype_counts.restype = C.c_int
    L.svt_genotype_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_void_p, C.c_int]
    L.svt_genotype.restype = C.c_int
    L.svt_genotype.argtypes = [C.POINTER(CEvidenceBatch), C.c_void_p, C.c_int, C.c_uint]
    if L.svt_version() != ABI_VERSION:
        raise SvtyperHipError("ABI mismatch: library %d, binding %d" % (L.svt_version(), ABI_VERSION))
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        msg = load().svt_last_error()
        raise SvtyperHipError("svtyper_hip error %d: %s" % (rc, msg.decode("utf-8", "replace") if msg else ""))


def trim():
    """Release the device scratch cached between batch creations (svt_trim)."""
    load().svt_trim()


def device_count() -> int:
    return int(load().svt_device_count())


ERR_UNSUPPORTED = -7


class PackedEvidence:
    """Packed evidence of a batch on the host (svt_packed_evidence, svt_pack_evidence): three sparse streams of small
    entries per unit instead of 16-byte records -- what crosses PCIe when the records were produced on the host."""

    def __init__(self, batch: EvidenceBatch):
        L = load()
        self._lib = L
        self._p = C.POINTER(CPackedEvidence)()
        cb = batch.as_c()
        _check(L.svt_pack_evidence(C.byref(cb), C.byref(self._p)))

    @classmethod
    def try_pack(cls, batch: EvidenceBatch):
        """None when the batch cannot be expressed as packed evidence (a histogram of more than 2047 bins, ...)."""
        try:
            return cls(batch)
        except SvtyperHipError as e:
            if "error %d" % ERR_UNSUPPORTED in str(e):
                return None
            raise

    @property
    def c(self) -> CPackedEvidence:
        return self._p.contents

    @property
    def n_units(self) -> int:
        return int(self.c.n_units)

    @property
    def n_records(self) -> int:
        return int(self.c.n_records)

    @property
    def nbytes(self) -> int:
        """bytes svt_batch_create_packed uploads: slots + slot offsets + unit headers"""
        return 16 * int(self.c.n_slots) + 4 * (3 * self.n_units + 1) + 16 * self.n_units

    def slots(self) -> np.ndarray:
        """uint32 [n_slots, 4] view of the slots"""
        n = int(self.c.n_slots)
        if n == 0:
            return np.zeros((0, 4), np.uint32)
        return np.ctypeslib.as_array(C.cast(self.c.slots, C.POINTER(C.c_uint32)), shape=(n, 4))

    def slot_offset(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.c.slot_offset, shape=(3 * self.n_units + 1,))

    def free(self):
        if self._p:
            self._lib.svt_packed_free(self._p)
            self._p = C.POINTER(CPackedEvidence)()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()


class _PinnedBuffer:
    def __init__(self, nbytes: int):
        self._lib = load()
        self.nbytes = max(int(nbytes), 1)
        self.ptr = self._lib.svt_pinned_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("svt_pinned_alloc(%d)" % self.nbytes)

    def __del__(self):
        try:
            if self.ptr:
                self._lib.svt_pinned_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_results(n_units: int) -> Results:
    """A Results whose records live in page-locked memory from the library's pool (svt_pinned_alloc).  The block is
    owned by the ARRAY (its base keeps the buffer object alive), so a caller that keeps only `.rec` -- or a view of
    it -- keeps the memory too; it returns to the pool when the last view is gone."""
    from .evidence import RESULT_DTYPE
    nbytes = int(n_units) * RESULT_DTYPE.itemsize
    buf = _PinnedBuffer(nbytes)
    raw = (C.c_uint8 * buf.nbytes).from_address(buf.ptr)
    raw._owner = buf           # ndarray.base -> this ctypes array -> the pooled block
    arr = np.frombuffer(raw, dtype=np.uint8, count=nbytes)
    res = Results.__new__(Results)
    res.rec = arr.view(RESULT_DTYPE)
    res.site_qual = None
    return res


def _check_out(out: Results, n_units: int) -> Results:
    """The C side writes n_units * 128 bytes through out.ptr() (by DMA when the block is page-locked)."""
    if out.n_units != int(n_units):
        raise ValueError("out holds %d records, the batch has %d units" % (out.n_units, int(n_units)))
    if not out.rec.flags["C_CONTIGUOUS"] or not out.rec.flags["WRITEABLE"]:
        raise ValueError("out.rec must be a writable C-contiguous array of result records")
    return out


def genotype_packed(packed: PackedEvidence, device: int = 0, flags: int = 0, out: Optional[Results] = None) -> Results:
    """svt_genotype_packed: create_packed + one pass + results + destroy, upload / pass / download overlapped by unit
    ranges.  `out`: a Results to fill (pinned_results(n): the records then come down by DMA while later pieces go up)."""
    out = Results.empty(packed.n_units) if out is None else _check_out(out, packed.n_units)
    _check(load().svt_genotype_packed(packed._p, C.c_void_p(out.ptr()), int(device), int(flags)))
    return out


def genotype_packed_from_records(batch: EvidenceBatch, device: int = 0, flags: int = 0, out: Optional[Results] = None) -> Results:
    """svt_genotype_packed_from_records: canonical records in host memory -> results through packed evidence, the host
    encoder running ahead of the upload by unit ranges (encode || upload || pass || download)."""
    out = Results.empty(batch.n_units) if out is None else _check_out(out, batch.n_units)
    cb = batch.as_c()
    _check(load().svt_genotype_packed_from_records(C.byref(cb), C.c_void_p(out.ptr()), int(device), int(flags)))
    return out


class DeviceBatch:
    """A batch resident in HBM (svt_batch)."""

    def __init__(self, batch: EvidenceBatch, device: int = 0, flags: int = 0):
        L = load()
        self._lib = L
        self._h = C.c_void_p()
        self.n_units = batch.n_units
        self.n_records = batch.n_records
        self.device = int(device)
        cb = batch.as_c()
        _check(L.svt_batch_create(C.byref(cb), int(device), int(flags), C.byref(self._h)))

    @classmethod
    def from_segments(cls, sbatch, device: int = 0, flags: int = 0):
        """svt_batch_create_segments: an evidence.SegmentedBatch -- the records go up piece by piece from where they lie
        (one reader per sample of a joint run), nothing is concatenated on the host.  The same batch as
        DeviceBatch(sbatch.joined())."""
        from .evidence import EvidenceBatch, RECORD_DTYPE
        L = load()
        self = cls.__new__(cls)
        self._lib = L
        self._h = C.c_void_p()
        self.n_units = sbatch.n_units
        self.n_records = sbatch.n_records
        self.device = int(device)
        head = EvidenceBatch.__new__(EvidenceBatch)      # (the unit arrays and libraries; `records` is ignored by the call)
        head.rec_offset, head.units, head.records = sbatch.rec_offset, sbatch.units, np.zeros(0, RECORD_DTYPE)
        head.libs, head.split_weight, head.disc_weight, head._keep = sbatch.libs, sbatch.split_weight, sbatch.disc_weight, []
        cb = head.as_c()
        segs = np.zeros(max(1, len(sbatch.segments)), np.dtype([("records", "<u8"), ("n_records", "<u8")]))
        for k, x in enumerate(sbatch.segments):
            segs[k] = (x.ctypes.data if x.shape[0] else 0, x.shape[0])
        _check(L.svt_batch_create_segments(C.byref(cb), segs.ctypes.data, len(sbatch.segments), int(device), int(flags), C.byref(self._h)))
        return self

    @classmethod
    def from_fragments(cls, fbatch, device: int = 0, flags: int = 0, return_records: bool = False):
        """Geometry on the device: fragment summaries (svtyper_amd.geometry.FragmentBatch) -> evidence
        records -> resident batch.  With return_records the derived canonical records come back too."""
        import numpy as np
        from .evidence import RECORD_DTYPE
        L = load()
        self = cls.__new__(cls)
        self._lib = L
        self._h = C.c_void_p()
        self.n_units = fbatch.n_units
        self.device = int(device)
        self.n_records = fbatch.n_fragments
        recs = np.zeros(fbatch.n_fragments, RECORD_DTYPE) if return_records else None
        cb = fbatch.as_c()
        _check(L.svt_batch_create_from_fragments(C.byref(cb), int(device), int(flags),
                                                 C.c_void_p(recs.ctypes.data) if return_records and recs.size else None,
                                                 C.byref(self._h)))
        self.records = recs
        return self

    @classmethod
    def from_packed(cls, packed: "PackedEvidence", device: int = 0, flags: int = 0):
        """svt_batch_create_packed: the slots go to HBM as they are."""
        L = load()
        self = cls.__new__(cls)
        self._lib = L
        self._h = C.c_void_p()
        self.n_units = packed.n_units
        self.device = int(device)
        self.n_records = packed.n_records
        _check(L.svt_batch_create_packed(packed._p, int(device), int(flags), C.byref(self._h)))
        return self

    def genotype(self, sync: bool = True):
        _check(self._lib.svt_batch_genotype(self._h, int(sync)))

    def synchronize(self):
        """svt_batch_sync: wait for the passes enqueued with genotype(sync=False) / genotype_n and report what they found."""
        _check(self._lib.svt_batch_sync(self._h))

    def genotype_n(self, iters: int):
        """Enqueue `iters` passes on the batch stream without waiting."""
        _check(self._lib.svt_batch_genotype_n(self._h, int(iters)))

    def genotype_timed(self, iters: int) -> float:
        """Elapsed milliseconds (HIP events on the batch stream) of `iters` passes."""
        ms = C.c_float()
        _check(self._lib.svt_batch_genotype_timed(self._h, int(iters), C.byref(ms)))
        return float(ms.value)

    def tune_placement(self, result_candidates: int = 32, record_candidates: int = 8) -> dict:
        """svt_batch_tune_placement: audition freshly allocated device buffers for the result records and the records with
        the real pass, keep the fastest (where a buffer lies in HBM moves the pass by up to 8 %).  Returns the pass time per
        launch before and after, in ms."""
        if any(r() is not None for r in getattr(self, "_views", ())):
            raise SvtyperHipError("tune_placement may replace the result buffer: drop the device_results_tensor() views first")
        before, after = C.c_float(), C.c_float()
        _check(self._lib.svt_batch_tune_placement(self._h, int(result_candidates), int(record_candidates), C.byref(before), C.byref(after)))
        return {"before_ms": float(before.value), "after_ms": float(after.value), "result_candidates": int(result_candidates),
                "record_candidates": int(record_candidates)}

    def results(self, out: Optional[Results] = None) -> Results:
        """The result records of the last pass.  `out`: a Results to fill instead of a fresh one -- pinned_results(n)
        gives one in page-locked memory, which the records reach by straight DMA."""
        out = Results.empty(self.n_units) if out is None else _check_out(out, self.n_units)
        _check(self._lib.svt_batch_results(self._h, C.c_void_p(out.ptr()), self.n_units))
        return out

    def result_order(self, n_samples: int):
        """svt_batch_result_order: the units are sample-major (unit = sample * n_sites + site); the pass writes their
        records site-major (index site * n_samples + sample).  0 / 1 = unit order."""
        _check(self._lib.svt_batch_result_order(self._h, int(n_samples)))

    def result_bytes(self) -> int:
        """bytes of one DEVICE result record: 128 (svt_result), or 96 under FLAG_RESULT96 (svt_result96)"""
        return int(self._lib.svt_batch_result_bytes(self._h))

    def result_slots(self) -> int:
        """records in the device result buffer: n_units, or under FLAG_RESULT96 the slots of the pass's workgroups (tagged
        records in the kernel's order, padding records included)"""
        return int(self._lib.svt_batch_result_slots(self._h))

    def device_results_ptr(self) -> int:
        """Device address of the svt_result[n_units] array the kernel writes to."""
        p = C.c_void_p()
        _check(self._lib.svt_batch_device_results(self._h, C.byref(p)))
        return int(p.value or 0)

    def device_results_tensor(self):
        """The batch's result records in HBM as a torch uint8 tensor: a zero-copy view of the buffer the pass writes, e.g. to
        hand to torch.distributed for the gather.  (The other way round -- binding a tensor torch allocated -- works too, but
        where the result records lie in HBM relative to the records decides 3-6 % of the pass time, and the batch's own
        buffer is the placement that measured fast: DESIGN.md 3.1.)

        Lifetime: the tensor's STORAGE owns a reference to this batch (through the array-interface object torch keeps as the
        storage's owner), so every derived view -- slices, .view(), what torch.distributed holds -- keeps the batch from
        being garbage-collected.  An explicit close() (or leaving the `with` block) while such a storage is alive raises
        instead of handing the buffer back to the pool under the view.  The device ordinal is the library's: torch and the
        library share one HIP runtime in the process and therefore one enumeration."""
        import weakref

        import torch
        from .evidence import RESULT_DTYPE

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (max(self.result_slots(), 1) * self.result_bytes(),), "typestr": "|u1",
                                      "data": (self.device_results_ptr(), False), "version": 2}
        v._svt_batch = self          # storage -> v -> batch
        t = torch.as_tensor(v, device=torch.device("cuda", getattr(self, "device", 0)))
        if not hasattr(self, "_views"):
            self._views = []
        self._views.append(weakref.ref(v))
        del v
        return t

    def bind_device_results(self, dev_ptr: int, capacity_bytes: Optional[int] = None):
        """Caller-owned device buffer of result_slots() * result_bytes() bytes, 128-byte aligned (n_units * 128 for 128-byte
        records; whole workgroups of tagged records under FLAG_RESULT96), e.g. a torch tensor's data_ptr(); 0 returns to the
        library's own buffer.  With `capacity_bytes` the library refuses a buffer that is too small."""
        if capacity_bytes is None or not dev_ptr:
            _check(self._lib.svt_batch_bind_device_results(self._h, C.c_void_p(int(dev_ptr) or None)))
        else:
            _check(self._lib.svt_batch_bind_device_results2(self._h, C.c_void_p(int(dev_ptr)), C.c_uint64(int(capacity_bytes))))

    def bytes(self):
        a, r = C.c_uint64(), C.c_uint64()
        _check(self._lib.svt_batch_bytes(self._h, C.byref(a), C.byref(r)))
        return int(a.value), int(r.value)

    def site_qual(self, n_samples: int, initial=None) -> np.ndarray:
        """QUAL of every site (classic.py:485,498) from the result records on the device; units must be
        site-major.  `initial`: incoming QUAL per site (--sum_quals), None = 0."""
        n_sites = self.n_units // max(1, int(n_samples))
        out = np.zeros(n_sites, np.float64)
        init = None if initial is None else np.ascontiguousarray(initial, dtype=np.float64)
        if init is not None and init.shape[0] != n_sites:
            raise ValueError("initial must have one value per site")
        _check(self._lib.svt_batch_site_qual(self._h, int(n_samples), None if init is None else init.ctypes.data,
                                             out.ctypes.data, n_sites))
        return out

    def table_mode(self) -> int:
        """0 one library, tables in LDS / 1 library windows in LDS (svt_unit.libs hints) / 2 general (tables through L2)"""
        c, m = C.c_int(), C.c_int()
        _check(self._lib.svt_batch_layout(self._h, C.byref(c), C.byref(m)))
        return int(m.value)

    def layout_name(self) -> str:
        """"stream" (the canonical CSR as uploaded) or "packed" (packed evidence as uploaded)"""
        c, m = C.c_int(), C.c_int()
        _check(self._lib.svt_batch_layout(self._h, C.byref(c), C.byref(m)))
        return {3: "stream", 4: "packed"}[c.value]

    def stream(self) -> int:
        return int(self._lib.svt_batch_stream(self._h) or 0)

    def close(self, force: bool = False):
        if self._h:
            if not force and any(r() is not None for r in getattr(self, "_views", ())):
                raise SvtyperHipError("DeviceBatch.close(): a tensor from device_results_tensor() (or a view of it) is still alive; "
                                      "drop it first -- its memory goes back to the buffer pool here")
            self._lib.svt_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close(force=exc[0] is not None)   # (never mask the exception that is leaving the block)


def expand96(records, n_units: int, out_rec=None) -> Results:
    """svt_results_expand96: tagged 96-byte device records (a uint8 / RESULT96_DTYPE array in host memory, e.g. one rank's part
    of a gather) -> `n_units` results in unit order, the eight derived counts restored (classic.py:455-469).  `out_rec`: a
    RESULT_DTYPE array (view) of n_units records to fill instead of a fresh Results."""
    from .evidence import RESULT96_DTYPE, RESULT_DTYPE
    a = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
    if a.size % RESULT96_DTYPE.itemsize:
        raise ValueError("not a whole number of 96-byte result records")
    n_rec = a.size // RESULT96_DTYPE.itemsize
    if out_rec is None:
        res = Results.empty(n_units)
        out_rec = res.rec
    else:
        res = None
        if out_rec.dtype != RESULT_DTYPE or out_rec.shape != (n_units,) or not out_rec.flags.c_contiguous:
            raise ValueError("out_rec must be a contiguous RESULT_DTYPE array of n_units records")
    _check(load().svt_results_expand96(C.c_void_p(a.ctypes.data), n_rec, C.c_void_p(out_rec.ctypes.data), int(n_units)))
    return res if res is not None else Results(out_rec)


def host_sq(results: Results) -> Results:
    """svt_results_host_sq: SQ of every called unit recomputed in place from the bit-exact GL with the host libm
    (the reference's own arithmetic, classic.py:473-481), so formatted SQ / QUAL are byte-identical."""
    _check(load().svt_results_host_sq(C.c_void_p(results.ptr()), results.n_units))
    return results


def site_qual_host(results: Results, n_samples: int, initial=None):
    """QUAL of every site (classic.py:216-217,485,498) from the result records on the host: the running binary64
    sum of SQ over the site's samples in order, reset by a blank sample.  Units site-major."""
    import numpy as np
    n_sites = results.n_units // max(1, int(n_samples))
    q = np.zeros(n_sites) if initial is None else np.array(initial, dtype=np.float64, copy=True)
    rec = results.rec[: n_sites * n_samples].reshape(n_sites, n_samples)
    for k in range(n_samples):
        gt = rec["gt"][:, k]
        q = np.where(gt >= 0, q + rec["sq"][:, k], np.where(gt == GT_BLANK, 0.0, q))
    return q


def _finish(d: "DeviceBatch", site_qual) -> Results:
    d.genotype(sync=True)
    res = host_sq(d.results())
    if site_qual is not None:      # (n_samples, initial QUAL per site or None): classic.py:485,498 over the refined SQ
        res.site_qual = site_qual_host(res, site_qual[0], site_qual[1])
    return res


def genotype_fragments(fbatch, device: int = 0, flags: int = 0, site_qual=None) -> Results:
    """Fragment summaries -> results with both the geometry and the likelihood stage on the device."""
    with DeviceBatch.from_fragments(fbatch, device, flags) as d:
        return _finish(d, site_qual)


def genotype_batch(batch: EvidenceBatch, device: int = 0, flags: int = 0, site_qual=None, out: Optional[Results] = None) -> Results:
    """create + genotype + results + destroy (svt_genotype: upload, pass and download overlapped by unit ranges).
    `out`: a Results to fill instead of a fresh one (pinned_results(n) for a page-locked one); not accepted together
    with `site_qual` (that route refines SQ on a fresh host copy)."""
    if site_qual is not None:
        if out is not None:
            raise ValueError("out= cannot be combined with site_qual=")
        with DeviceBatch(batch, device, flags) as d:
            return _finish(d, site_qual)
    L = load()
    out = Results.empty(batch.n_units) if out is None else _check_out(out, batch.n_units)
    cb = batch.as_c()
    _check(L.svt_genotype(C.byref(cb), C.c_void_p(out.ptr()), int(device), int(flags)))
    return out


FORMAT_CODES = {name: i for i, name in enumerate(
    ("GT", "GQ", "SQ", "GL", "DP", "RO", "AO", "QR", "QA", "RS", "AS", "ASC", "RP", "AP", "AB"))}
FORMAT_ABSENT = 255


def format_results(results: Results, fields, skipped_as_dots: bool):
    """svt_format_results: the text of every unit's VCF sample column for the FORMAT keys `fields` (names; a
    key outside the fifteen svtyper ones prints '.'), as a list of str.  Host-only (no GPU needed)."""
    L = load()
    codes = np.array([FORMAT_CODES.get(f, FORMAT_ABSENT) for f in fields], dtype=np.uint8)
    n = results.n_units
    text, off = C.c_void_p(), C.c_void_p()
    _check(L.svt_format_results(C.c_void_p(results.ptr()), n, codes.ctypes.data, len(codes), 1 if skipped_as_dots else 0,
                                C.byref(text), C.byref(off)))
    try:
        o = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), shape=(n + 1,)).tolist()
        blob = C.string_at(text, o[-1]).decode("ascii")
        return [blob[o[i]:o[i + 1]] for i in range(n)]
    finally:
        L.svt_format_free(text, off)


def bayes_gt_array(ref, alt, is_dup, device: int = 0):
    """svt_bayes_gt: arrays of (ref, alt, is_dup) -> float64 [n, 4] = (lp_homref, lp_het, lp_homalt,
    log_choose(ref + alt, alt)), evaluated on the device (svtyper/statistics.py:9-37)."""
    import numpy as np
    L = load()
    r = np.ascontiguousarray(ref, dtype=np.int32)
    a = np.ascontiguousarray(alt, dtype=np.int32)
    d = np.ascontiguousarray(np.broadcast_to(np.asarray(is_dup, dtype=np.uint8), r.shape))
    if r.shape != a.shape:
        raise ValueError("ref and alt must have the same shape")
    out = np.zeros((r.size, 4), np.float64)
    _check(L.svt_bayes_gt(r.ctypes.data, a.ctypes.data, d.ctypes.data, r.size, out.ctypes.data, int(device)))
    return out


def genotype_counts(counts, is_dup, split_weight: float = 1.0, disc_weight: float = 1.0, device: int = 0) -> Results:
    """svt_genotype_counts: the array form of the reference's bayesian_genotype(breakpoint, counts, ...)
    (svtyper/singlesample.py:406-473).  `counts`: float64 [n, 5] = (ref_seq, alt_seq, alt_clip, ref_span, alt_span)
    as tally_variant_read_fragments returned them; `is_dup`: [n].  No zeroing rule, no blank result."""
    import numpy as np
    L = load()
    c = np.ascontiguousarray(counts, dtype=np.float64).reshape(-1, 5)
    d = np.ascontiguousarray(np.broadcast_to(np.asarray(is_dup, dtype=np.uint8), (c.shape[0],)))
    out = Results.empty(c.shape[0])
    _check(L.svt_genotype_counts(c.ctypes.data, d.ctypes.data, c.shape[0], float(split_weight), float(disc_weight),
                                 C.c_void_p(out.ptr()), int(device)))
    return out


def genotype_multi(batch: EvidenceBatch, devices, group: int = 1, flags: int = 0) -> Results:
    """svt_genotype_multi: `batch` sharded over the listed devices (one host thread per entry, a device may be
    listed more than once), result records in unit order."""
    L = load()
    out = Results.empty(batch.n_units)
    cb = batch.as_c()
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    _check(L.svt_genotype_multi(C.byref(cb), C.c_void_p(out.ptr()), devs, len(devices), int(group), int(flags)))
    return out


def shard_bounds(rec_offset, n_shards: int, group: int = 1):
    """svt_shard_bounds: [(lo, hi)] unit ranges, one per shard (host only)."""
    import numpy as np
    L = load()
    off = np.ascontiguousarray(rec_offset, dtype=np.uint64)
    b = np.zeros(n_shards + 1, np.uint64)
    _check(L.svt_shard_bounds(off.ctypes.data, max(0, off.shape[0] - 1), int(n_shards), int(group), b.ctypes.data))
    return [(int(b[i]), int(b[i + 1])) for i in range(n_shards)]


def chunk_bounds(rec_offset, group: int = 1, max_records: int = 0):
    """svt_chunk_bounds: [(lo, hi)] unit ranges of the fewest chunks that each fit one resident batch (at most `max_records`
    records, 0 = the library's 32-bit bound), cut at multiples of `group` units (samples per site).  Host only.  A batch
    beyond the bound goes through DeviceBatch chunk by chunk (`batch.slice(lo, hi)`); genotype_batch / svt_genotype do it
    themselves."""
    import numpy as np
    L = load()
    off = np.ascontiguousarray(rec_offset, dtype=np.uint64)
    n = max(0, off.shape[0] - 1)
    count = C.c_uint32()
    _check(L.svt_chunk_bounds(off.ctypes.data, n, int(group), int(max_records), None, 0, C.byref(count)))
    b = np.zeros(count.value + 1, np.uint64)
    _check(L.svt_chunk_bounds(off.ctypes.data, n, int(group), int(max_records), b.ctypes.data, count.value, C.byref(count)))
    return [(int(b[i]), int(b[i + 1])) for i in range(count.value)]
