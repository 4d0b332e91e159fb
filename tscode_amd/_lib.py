"""ctypes binding of libtscode_hip.so (include/tscode_hip.h).

There is no fallback: if the shared library is missing, or no gfx950 device is usable, the
product fails loudly (``TscodeHipError``).  Nothing in this package imports the CPU oracle.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSCODE_AMD_LIB") or os.path.join(_HERE, "libtscode_hip.so")   # override: A/B builds

TSC_MAX_PASSES = 18

c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)
c_f32p = C.POINTER(C.c_float)


class TscodeHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libtscode_hip error {code}: {message}")
        self.code = code


class PassStats(C.Structure):
    _fields_ = [("k", C.c_int64), ("n_active_before", C.c_int64), ("n_active_after", C.c_int64),
                ("pairs_evaluated", C.c_int64), ("pairs_computed", C.c_int64), ("candidates", C.c_int64),
                ("pairs_screened", C.c_int64), ("new_keys", C.c_int64), ("gpu_ms", C.c_double), ("tile_ms", C.c_double),
                ("algo", C.c_int32), ("nonfinite_input", C.c_int32)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class ExchangeRecord(C.Structure):
    """tsc_exchange_record: one collective of tsc_prune_run_sharded (k < 0: the cache views in front of pass -k)."""
    _fields_ = [("k", C.c_int64), ("kind", C.c_int32), ("count", C.c_int64)]


XCHG_SUM_I64, XCHG_MIN_I32 = 1, 2
XCHG_HANDLE_BYTES = 64
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64)     # tsc_exchange_fn(user, kind, buf_dev, count)

# name -> (restype, argtypes); mirrors include/tscode_hip.h one to one
_vp = C.c_void_p
_SIGNATURES = {
    "tsc_version": (C.c_int, []),
    "tsc_last_error": (C.c_char_p, []),
    "tsc_build_digest": (C.c_char_p, []),
    "tsc_device_count": (C.c_int, []),
    "tsc_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "tsc_ctx_destroy": (C.c_int, [_vp]),
    "tsc_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "tsc_ctx_synchronize": (C.c_int, [_vp]),
    "tsc_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "tsc_malloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "tsc_free": (C.c_int, [_vp, _vp]),
    "tsc_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "tsc_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "tsc_timer_begin": (C.c_int, [_vp]),
    "tsc_timer_end": (C.c_int, [_vp, c_f32p]),
    "tsc_transform_batch": (C.c_int, [_vp, _vp, c_i64p, c_i32p, c_i32p, C.c_int, _vp, _vp, _vp, C.c_int64, _vp]),
    "tsc_transform_batch_dev": (C.c_int, [_vp, _vp, c_i64p, c_i32p, c_i32p, C.c_int, _vp, _vp, _vp, C.c_int64, _vp]),
    "tsc_clash_mask": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, c_i32p, C.c_int, C.c_double, C.c_int64, _vp, _vp]),
    "tsc_clash_mask_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, c_i32p, C.c_int, C.c_double, C.c_int64, _vp, _vp]),
    "tsc_embed_clash_mask_dev": (C.c_int, [_vp, _vp, c_i64p, c_i32p, c_i32p, C.c_int, _vp, _vp, _vp, C.c_int64, C.c_double,
                                           C.c_int64, _vp, _vp]),
    "tsc_all_dists": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp]),
    "tsc_compact_rows_dev": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int64, _vp, c_i64p]),
    "tsc_gather_heavy_dev": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int, c_i32p, C.c_int, _vp, c_i64p]),
    "tsc_rmsd_pairs": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int64, _vp, _vp]),
    "tsc_screen_mm_values": (C.c_int, [_vp, _vp, C.c_int64, C.c_double, _vp, _vp, _vp]),
    "tsc_rmsd_pairs_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int64, _vp, _vp]),
    "tsc_embed_clash_compact_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int64, _vp, C.c_int, C.c_double, C.c_int64,
                                               _vp, _vp, _vp, _vp]),
    "tsc_basis_from_poses_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int64, _vp, C.c_int]),
    "tsc_embed_masked_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int64, _vp, _vp, C.c_int, _vp, _vp, c_i64p]),
    "tsc_inertia_moments": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp]),
    "tsc_moi_first_similar": (C.c_int, [_vp, _vp, C.c_int64, C.c_double, _vp]),
    "tsc_embed_scores": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, C.c_int, _vp, _vp]),
    "tsc_torsion_fingerprints": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, _vp]),
    "tsc_tfd_first_similar": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double, _vp]),
    "tsc_string_embed_params": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int, _vp, _vp, _vp]),
    "tsc_string_embed_params_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int, _vp, _vp, _vp]),
    "tsc_cyclical_embed_params": (C.c_int, [_vp] * 10 + [C.c_int64, _vp, _vp]),
    "tsc_csearch_rotate": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int64, C.c_double, C.c_int64, _vp, _vp]),
    "tsc_csearch_rotate_dev": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int64, C.c_double, C.c_int64, _vp, _vp]),
    "tsc_torsion_comp_check": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, C.c_double, C.c_int64, _vp]),
    "tsc_rotate_dihedral": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp]),
    "tsc_greedy_group_filter": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_double, _vp]),
    "tsc_greedy_group_filter_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, C.c_int, C.c_double, _vp]),
    "tsc_tfd_greedy_filter": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_double, _vp, c_i64p]),
    "tsc_string_embed": (C.c_int, [_vp, _vp, c_i64p, c_i32p, c_i32p, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int, C.c_double, C.c_int64,
                                   _vp, C.c_int, C.c_double, _vp, _vp, _vp, C.c_int64, c_i64p, c_i64p]),
    "tsc_cyclical_embed": (C.c_int, [_vp, _vp, c_i64p, c_i32p, c_i32p, C.c_int] + [_vp] * 10 + [C.c_int64, _vp, C.c_int, C.c_double, C.c_int64,
                                     C.c_double, _vp, _vp, _vp, C.c_int64, c_i64p, c_i64p]),
    "tsc_host_graph_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int64, _vp]),
    "tsc_prune_rmsd": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_double, C.c_int, _vp, C.POINTER(PassStats), C.POINTER(C.c_int)]),
    "tsc_prune_structures": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, c_i32p, C.c_int, C.c_double, C.c_int, _vp, C.POINTER(PassStats),
                                       C.POINTER(C.c_int)]),
    "tsc_prune_rmsd_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_double, C.c_int, _vp, C.POINTER(PassStats), C.POINTER(C.c_int)]),
    "tsc_prune_create": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_double, C.c_int, C.POINTER(_vp)]),
    "tsc_prune_next_pass": (C.c_int, [_vp, c_i64p]),
    "tsc_prune_pass_estimate": (C.c_int, [_vp, c_i64p]),
    "tsc_prune_run_replicated": (C.c_int, [_vp, C.c_int, C.c_int64, c_i64p]),
    "tsc_prune_pass_local": (C.c_int, [_vp, C.c_int, C.c_int]),
    "tsc_prune_pass_rows": (C.c_int, [_vp, C.c_int, C.c_int]),
    "tsc_prune_exchange_words": (C.c_int, [C.c_int64, C.c_int, c_i64p]),
    "tsc_prune_set_partition": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int64]),
    "tsc_prune_pass_partitioned": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "tsc_prune_pass_range": (C.c_int, [_vp]),
    "tsc_prune_pass_merge": (C.c_int, [_vp]),
    "tsc_prune_views_ptr": (C.c_int, [_vp, C.POINTER(_vp), c_i64p, c_i64p]),
    "tsc_prune_views_merged": (C.c_int, [_vp]),
    "tsc_prune_best_ptr": (C.c_int, [_vp, C.POINTER(_vp), c_i64p]),
    "tsc_prune_use_best_buffer": (C.c_int, [_vp, _vp]),
    "tsc_prune_pass_finish": (C.c_int, [_vp]),
    "tsc_prune_mask_dev": (C.c_int, [_vp, C.POINTER(_vp)]),
    "tsc_prune_run_sharded": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int64, _vp, C.c_int64, EXCHANGE_FN, _vp, C.POINTER(ExchangeRecord), C.c_int,
                                        C.POINTER(C.c_int)]),
    "tsc_xchg_slot_bytes": (C.c_int, [C.c_int64, C.c_int, c_i64p]),
    "tsc_xchg_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int64, C.POINTER(_vp), _vp]),
    "tsc_xchg_connect": (C.c_int, [_vp, _vp]),
    "tsc_xchg_set_timeout": (C.c_int, [_vp, C.c_double]),
    "tsc_xchg_allreduce": (C.c_int, [_vp, C.c_int, _vp, C.c_int64]),
    "tsc_xchg_status": (C.c_int, [_vp, c_i64p, C.POINTER(C.c_int)]),
    "tsc_xchg_destroy": (C.c_int, [_vp]),
    "tsc_prune_copy_mask_dev": (C.c_int, [_vp, _vp]),
    "tsc_prune_stats": (C.c_int, [_vp, C.POINTER(PassStats), C.POINTER(C.c_int)]),
    "tsc_prune_destroy": (C.c_int, [_vp]),
    "tsc_pipeline_dev": (C.c_int, [_vp, _vp, c_i64p, c_i32p, c_i32p, C.c_int, _vp, _vp, _vp, C.c_int64, c_i32p, C.c_int,
                                   C.c_double, C.c_int64, C.c_double, C.c_int, _vp, _vp, _vp, _vp, c_i64p, c_i64p,
                                   C.POINTER(PassStats), C.POINTER(C.c_int), c_f32p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None
_lock = threading.Lock()


def load():
    """Load libtscode_hip.so and declare every prototype.  Raises if the library is not built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise TscodeHipError(-2, f"{LIB_PATH} not found: build it with `python -m tscode_amd.build` "
                                         "(hipcc --offload-arch=gfx950); there is no CPU fallback")
            # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7, and the
            # copy that is loaded first serves every later NEEDED entry of that soname.  If this library
            # came first it would pull in /opt/rocm's runtime and torch would then find "no HIP GPUs";
            # importing torch first (it does not touch the GPU) makes both share torch's copy.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
            lib = C.CDLL(LIB_PATH)
            lax = bool(os.environ.get("TSCODE_AMD_LAX"))   # A/B runs against an older build (TSCODE_AMD_LIB): symbols it lacks are skipped
            for name, (res, args) in _SIGNATURES.items():
                if lax and not hasattr(lib, name):
                    continue
                fn = getattr(lib, name)          # AttributeError here = header/library mismatch
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise TscodeHipError(rc, load().tsc_last_error().decode(errors="replace"))


def ptr(a):
    """Device pointer of a torch tensor / raw int, or host pointer of a contiguous NumPy array."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(a.data_ptr())      # torch.Tensor
