"""The reference's inner operator seams `statistics.log_choose` / `statistics.bayes_gt`
(svtyper/statistics.py:9-37), evaluated on the MI355X through `svt_bayes_gt`.

Same names, argument meaning and return shapes as the reference; the array forms are the ones a
batch-oriented caller should use.  No CPU implementation lives here.
"""
from __future__ import annotations

from . import hip


def bayes_gt(ref, alt, is_dup, device: int = 0):
    """(lp_homref, lp_het, lp_homalt), log10 scaled (statistics.py:23-37)."""
    out = hip.bayes_gt_array([int(ref)], [int(alt)], [bool(is_dup)], device)
    return (float(out[0, 0]), float(out[0, 1]), float(out[0, 2]))


def log_choose(n, k, device: int = 0):
    """log10 of C(n, k) by the reference's k-term loop (statistics.py:9-20)."""
    n, k = int(n), int(k)
    return float(hip.bayes_gt_array([n - k], [k], [False], device)[0, 3])


def bayes_gt_array(ref, alt, is_dup, device: int = 0):
    """float64 [n, 3] of log10 likelihoods for arrays of (ref, alt, is_dup)."""
    return hip.bayes_gt_array(ref, alt, is_dup, device)[:, :3]
