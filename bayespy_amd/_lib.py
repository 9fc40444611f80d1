"""ctypes binding of libdme_hip.so (include/dme_hip.h).  No fallback: if the library is missing
or no GPU is present, every entry point of the package fails loudly."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdme_hip.so")

OK = 0
ERR_INVALID_ARG = 1
ERR_HIP = 2
ERR_UNSUPPORTED = 3
ERR_WORKSPACE = 4
ERR_ZERO_DIVISION = 5
ERR_MATH_DOMAIN = 6
ERR_INTERNAL = 7
PARSE_FALLBACK = 100

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_f64 = ctypes.c_double
_int = ctypes.c_int
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/dme_hip.h one to one
SIGNATURES = {
    "dme_abi_version": (_int, []),
    "dme_last_error": (ctypes.c_char_p, []),
    "dme_device_cus": (_int, []),
    "dme_set_cu_limit": (_int, [_int]),
    "dme_stream_create_masked": (_int, [_vp, _int, _vp]),
    "dme_stream_destroy": (_int, [_vp]),
    "dme_stats_workspace_bytes": (_i64, [_i64, _i64]),
    "dme_overflow_bytes": (_i64, [_i64]),
    "dme_overflow_init": (_int, [_vp, _i64, _vp]),
    "dme_overflow_extents": (_int, [_vp, _i64, _vp, _vp]),
    "dme_overflow_append": (_int, [_vp, _vp, _vp, _vp]),
    "dme_stats_dense_u16": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dme_stats_dense_i32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dme_bounds_dense_u16": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "dme_bounds_dense_i32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "dme_bounds_csr_i32": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "dme_dense_row_nnz_u16": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "dme_dense_row_nnz_i32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "dme_dense_to_csr_u16": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "dme_dense_to_csr_i32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "dme_stats_csr_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "dme_stats_csr_i32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp,
                                 _i64, _vp]),
    "dme_suffix_scan": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "dme_ragged_plan": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "dme_suffix_scan_ragged": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dme_stats_dense_weighted_i32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "dme_stats_dense_weighted_batch_i32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp,
                                                  _vp, _vp, _vp]),
    "dme_stats_weighted_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "dme_stats_dense_weighted_batch_ws_i32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp,
                                                     _vp, _vp, _vp, _i64, _vp]),
    "dme_suffix_scan_weighted": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "dme_suffix_scan_weighted_batch": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64,
                                              _vp, _vp]),
    "dme_cli_init_alphas": (_int, [_vp, _i64, _vp, _vp]),
    "dme_shard_message_words": (_i64, [_i64, _i64, _i64, _i64]),
    "dme_shard_pack": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "dme_shard_unpack": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "dme_shard_extents": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "dme_shard_offsets": (_int, [_vp, _i64, _vp, _vp]),
    "dme_shard_compact": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp]),
    "dme_shard_expand": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "dme_shard_message_words32": (_i64, [_i64, _i64, _i64, _i64]),
    "dme_shard_pack32": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "dme_shard_unpack32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "dme_solver_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "dme_eval": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _f64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dme_eval_noalloc": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _f64, _vp, _vp, _vp, _vp]),
    "dme_predicted_step": (_int, [_f64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "dme_predicted_step_alt": (_int, [_f64, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "dme_solve": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _f64, _vp, _vp, _i64, _vp,
                         _vp, _vp, _vp, _i64, _vp]),
    "dme_solve_enqueue": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _f64, _vp, _vp, _vp, _vp,
                                 _i64, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "dme_solve_finish": (_int, [_vp, _i64, _i64]),
    "dme_solver_workspace_bytes_ragged": (_i64, [_i64, _i64, _i64]),
    "dme_eval_ragged": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _f64, _vp, _f64, _int, _vp, _vp, _vp,
                               _vp, _i64, _vp]),
    "dme_solve_ragged": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _f64, _vp, _f64, _int, _vp,
                                _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dme_solve_enqueue_ragged": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _f64, _vp, _vp, _vp,
                                        _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "dme_solve_set_path": (_int, [_int]),
    "dme_parse_count_rows": (_i64, [ctypes.c_char_p, _i64]),
    "dme_parse_counts": (_int, [ctypes.c_char_p, _i64, _int, _i64, _vp, _i64, _vp, _int]),
    "dme_parse_counts_selected": (_int, [ctypes.c_char_p, _i64, _int, _i64, _vp, _i64, _vp, _i64, _vp, _int]),
    "dme_cap_draws_needed": (_i64, [_vp, _i64, _i64, _i64]),
    "dme_sample_and_cap_i32": (_int, [_vp, _i64, _i64, _f64, _i64, _vp, _i64, _vp, _vp]),
    "dme_eval_cpt": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _f64, _int, _vp, _vp, _vp,
                            _vp, _i64, _vp]),
    "dme_solve_cpt": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _f64, _int, _vp,
                             _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dme_synth_scratch_bytes": (_i64, [_i64]),
    "dme_synth_dirmult": (_int, [_vp, _int, _i64, _i64, _i64, _vp, _i64, _vp, _u64, _i64, _vp, _i64, _vp]),
}



class Ragged(ctypes.Structure):
    """dme_ragged_t (include/dme_hip.h): a host struct of device pointers."""
    _fields_ = [("offU", _vp), ("slotoff", _vp), ("slotcol", _vp), ("nslots", _i64), ("total", _i64)]


SEGMENT_TERMS = 256  # DME_SEGMENT_TERMS
ABI_VERSION = 2      # DME_ABI_VERSION

_lib = None


class DmeError(RuntimeError):
    """A failure inside libdme_hip.so that has no Python-exception counterpart in the reference."""


def lib():
    """Loads libdme_hip.so (built by `make -C bayespy_amd/csrc` or __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C bayespy_amd/csrc` "
                "(hipcc, gfx950).  bayespy_amd has no CPU fallback.")
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).
        # It must be in the process BEFORE this library is mapped, so that our DT_NEEDED
        # libamdhip64.so.7 binds to that same copy; two HIP runtimes in one process cannot both
        # see the GPU (the second reports "no ROCm-capable device") and could not share the
        # device pointers torch hands us anyway.
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.dme_abi_version() != ABI_VERSION:
            raise ImportError("libdme_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc):
    """Maps a DME_ERR_* code to the exception the reference would raise for the same condition."""
    if rc == OK:
        return
    msg = lib().dme_last_error().decode("utf-8", "replace")
    if rc == ERR_ZERO_DIVISION:
        raise ZeroDivisionError("float division by zero")  # DE/dme.py:69-71
    if rc == ERR_MATH_DOMAIN:
        raise ValueError("math domain error")  # math.log(x <= 0), DE/dme.py:21
    if rc == ERR_INVALID_ARG:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise DmeError(f"libdme_hip error {rc}: {msg}")
