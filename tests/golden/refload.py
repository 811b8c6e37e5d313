"""Load the (Python 2.7) reference svtyper package read-only, in memory, under Python 3.

DEV-CONTAINER ONLY: used by tests/golden/make_golden.py to generate golden vectors.
Nothing under tests/ imports this at test time and it never travels to the GPU box
(/root/reference does not exist there).  No reference source is copied: modules are
read from /root/reference, converted by the stdlib lib2to3 in memory and exec'd.

Shims:
  * lib2to3 fixes the Py2-only syntax (print statement, `except X, e`, `lambda(x)`,
    xrange, map/keys/values list-wrapping);
  * `pysam` and `cytoolz` are stand-in modules (pysam: whatever object the caller
    passes as `pysam_module`, e.g. svtyper_amd.bam's reader; cytoolz: partition_all);
  * SamFragment.p_concordant returns False where Python 2 evaluated `None > 0.5`.
"""
from __future__ import annotations

import os
import sys
import types
import warnings

REFERENCE_ROOT = os.environ.get("SVTYPER_REFERENCE", "/root/reference")


def _partition_all(n, seq):
    seq = list(seq)
    for i in range(0, len(seq), n):
        yield tuple(seq[i:i + n])


def load_reference(pysam_module=None):
    """Returns a namespace with .statistics .utils .parsers .classic .singlesample"""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor

        fixers = refactor.get_fixers_from_package("lib2to3.fixes")
        tool = refactor.RefactoringTool(fixers)

    pkg_dir = os.path.join(REFERENCE_ROOT, "svtyper")
    if not os.path.isdir(pkg_dir):
        raise RuntimeError("reference not found at %s (dev container only)" % REFERENCE_ROOT)

    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "svtyper" or
             k.startswith("svtyper.") or k in ("pysam", "cytoolz", "cytoolz.itertoolz")}
    pkg = types.ModuleType("svtyper")
    pkg.__path__ = []  # mark as package
    sys.modules["svtyper"] = pkg

    if pysam_module is None:
        pysam_module = types.ModuleType("pysam")
    sys.modules["pysam"] = pysam_module
    cyt = types.ModuleType("cytoolz")
    cyti = types.ModuleType("cytoolz.itertoolz")
    cyti.partition_all = _partition_all
    cyt.itertoolz = cyti
    sys.modules["cytoolz"] = cyt
    sys.modules["cytoolz.itertoolz"] = cyti

    ns = types.SimpleNamespace()
    try:
        for name in ("version", "statistics", "parsers", "utils", "classic", "singlesample"):
            path = os.path.join(pkg_dir, name + ".py")
            with open(path) as f:
                src = f.read()
            if not src.endswith("\n"):
                src += "\n"
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                src3 = str(tool.refactor_string(src, path))
            mod = types.ModuleType("svtyper." + name)
            mod.__file__ = path
            sys.modules["svtyper." + name] = mod
            setattr(pkg, name, mod)
            exec(compile(src3, path, "exec"), mod.__dict__)
            setattr(ns, name, mod)

        # Python 2: `None > 0.5` is False (parsers.py:879-882)
        orig = ns.parsers.SamFragment.p_concordant

        def p_concordant(self, var_length=None, _orig=orig):
            try:
                return _orig(self, var_length)
            except TypeError:
                return False

        ns.parsers.SamFragment.p_concordant = p_concordant
    finally:
        # leave the svtyper.* modules importable for the lifetime of `ns` only through ns;
        # restore anything we displaced
        for k in ("svtyper", "pysam", "cytoolz", "cytoolz.itertoolz"):
            if saved.get(k) is not None:
                sys.modules[k] = saved[k]
    return ns
