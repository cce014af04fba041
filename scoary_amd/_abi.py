"""ctypes binding of the C-ABI declared in include/scoary_hip.h.

The HIP library is the product path: if ``libscoary_hip.so`` is missing or the
GPU is absent, everything here raises -- there is no CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCOARY_HIP_LIB: an alternative build of the same sources (kernel A/B experiments, tools/ab_lib.sh)
LIB_PATH = os.environ.get("SCOARY_HIP_LIB") or os.path.join(_HERE, "csrc", "libscoary_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "scoary_hip.h")

ABI_VERSION = 9

_i64, _u64, _i32, _vp, _cp = (ctypes.c_int64, ctypes.c_uint64, ctypes.c_int,
                              ctypes.c_void_p, ctypes.c_char_p)

# name -> (restype, argtypes); mirrors include/scoary_hip.h one to one
SIGNATURES = {
    "scoary_abi_version": (_i32, []),
    "scoary_create": (_i32, [_i32, ctypes.POINTER(_vp)]),
    "scoary_destroy": (None, [_vp]),
    "scoary_last_error": (_cp, [_vp]),
    "scoary_tiled_quads": (_i64, [_i64]),
    "scoary_tiled_genes": (_i64, [_i64]),
    "scoary_tiled_bytes": (_i64, [_i64, _i64]),
    "scoary_row_words": (_i64, [_i64]),
    "scoary_pack_dense": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "scoary_tile_rows": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "scoary_counts": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "scoary_counts_traits_per_pass": (_i64, [_i64]),
    "scoary_trait_plan_bytes": (_i64, [_i64, _i64]),
    "scoary_trait_plan": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "scoary_counts_planned": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "scoary_fisher": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "scoary_fisher_lists": (_i32, [_vp, _vp, _i64, _i64] + [_vp] * 7),
    "scoary_fisher_scipy": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "scoary_fisher_scipy_max_isolates": (_i64, []),
    "scoary_perm_generate": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _u64, _vp, _vp]),
    "scoary_permute": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "scoary_permute_seq": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp,
                                  _vp]),
    "scoary_list_tiles_words": (_i64, [_i64, _i64, _i64]),
    "scoary_list_tile_words": (_i64, [_i64]),
    "scoary_list_params": (_i32, [_i64, _vp]),
    "scoary_list_max_isolates": (_i64, []),
    "scoary_list_segments": (_i64, [_i64]),
    "scoary_perm_generate_tiles": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _u64, _vp,
                                          _vp]),
    "scoary_perm_generate_tiles_range": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _u64, _i64,
                                                _i64, _vp, _vp]),
    "scoary_perm_max_isolates": (_i64, []),
    "scoary_permute_lists_scratch_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "scoary_permute_lists": (_i32, [_vp, _vp, _vp, _i64] + [_vp] * 8 + [_i64, _i64, _i64, _i64, _vp,
                                                                       _i32, _vp]),
    "scoary_lists_scratch_bytes": (_i64, [_i64, _i64]),
    "scoary_lists_slack_entries": (_i64, []),
    "scoary_lists_plan": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp,
                                 ctypes.POINTER(_i64), _vp]),
    "scoary_lists_fill": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp]),
    "scoary_hamming": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "scoary_upgma_scratch_bytes": (_i64, [_i64]),
    "scoary_upgma": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "scoary_gather_bits": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "scoary_tree_pairs": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "scoary_tree_permute": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp,
                                   _vp]),
    "scoary_row_hash": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "scoary_pack_records": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "scoary_gather": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "scoary_graph_begin": (_i32, [_vp, _vp]),
    "scoary_graph_end": (_i32, [_vp, _vp, ctypes.POINTER(_vp)]),
    "scoary_graph_launch": (_i32, [_vp, _vp, _vp]),
    "scoary_graph_destroy": (None, [_vp]),
    "scoary_set_timing": (_i32, [_vp, _i32]),
    "scoary_last_kernel_ms": (_i32, [_vp, _cp, ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


class ScoaryHipError(RuntimeError):
    pass


def load():
    """Load libscoary_hip.so (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScoaryHipError(
            "HIP extension not built: %s is missing. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "scoary_amd has no CPU fallback." % LIB_PATH)
    try:
        import torch  # noqa: F401  (loads libamdhip64 first; our .so binds to it by SONAME)
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    got = lib.scoary_abi_version()
    if got != ABI_VERSION:
        raise ScoaryHipError("ABI version mismatch: library %d, binding %d" % (got, ABI_VERSION))
    _lib = lib
    return lib
