"""svtyper_amd -- MI355X-native (gfx950) SVTyper likelihood hot path.

Host side mirrors the reference's Python surface (``svtyper_amd.classic.sv_genotype``,
``svtyper_amd.singlesample.sso_genotype``); the per-(breakpoint, sample) evidence tally,
``bayes_gt`` likelihood and GT/GQ/SQ decision run in hand-written HIP kernels behind the
C ABI of ``include/svtyper_hip.h`` (``svtyper_amd/csrc``).  There is no CPU fallback.
"""
__version__ = "0.1.0"
