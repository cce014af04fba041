"""scoary_amd -- MI355X-native association engine for Scoary's Fisher /
permutation hot path (Setup_results / Perform_statistics / --permute).

Host side mirrors the reference's interface for this path (scoary_amd.methods);
all arithmetic of the path runs in hand-written gfx950 HIP kernels behind the
C-ABI in include/scoary_hip.h.  There is no CPU fallback.
"""
__version__ = "0.1.0"
SCOARY_COMPAT_VERSION = "1.6.16"
