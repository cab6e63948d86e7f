"""ctypes binding of libpmn_hip.so (C ABI declared in include/pmn_hip.h).

The library is built in-tree (``patchmatchnet_amd/csrc/libpmn_hip.so``) by ``build()`` / ``make -C patchmatchnet_amd/csrc``
so that it travels with the source snapshot.  There is no fallback: if the library is missing or a symbol is absent
the product raises, it never silently runs something else.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "libpmn_hip.so")
ABI_VERSION = 22
MLP_FLOATS = 340
MAX_DEPTH = 64
MAX_NEIGHBORS = 17
MAX_FUSE_SRC = 32

_fp = ctypes.c_void_p  # device float* (passed as integer address)
_ip = ctypes.c_void_p
_hp = ctypes.c_void_p  # host pointer
_i = ctypes.c_int
_f = ctypes.c_float
_s = ctypes.c_void_p   # hipStream_t

# name -> argtypes; mirrors include/pmn_hip.h one to one (tests/test_abi.py checks the header against this table)
SIGNATURES = {
    "pmn_abi_version": [],
    "pmn_error_string": [_i],
    "pmn_nchw_to_nhwc": [_fp, _fp, _i, _i, _i, _i, _s],
    "pmn_feature_weight": [_fp, _fp, _hp, _fp, _i, _i, _i, _i, _i, _i, _fp, _s],
    "pmn_init_hypotheses": [_fp, _fp, _i, _fp, _fp, _i, _f, _fp, _hp, _i, _i, _i, _i, _fp, _fp, _s],
    "pmn_warp_correlate": [_fp, _fp, _fp, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp, _fp, _ip,
                           _fp, _s],
    "pmn_warp_correlate_views": [_fp, _fp, _fp, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp, _fp, _ip,
                                 _fp, _s],
    "pmn_aggregate_regress": [_fp, _fp, _fp, _fp, _fp, _hp, _i, _f, _i, _i, _i, _i, _i, _fp, _fp, _s],
    "pmn_confidence": [_fp, _i, _i, _i, _i, _i, _i, _fp, _ip, _s],
    "pmn_conv2d": [_fp, _fp, _fp, _fp, _fp] + [_i] * 14 + [_s],
    "pmn_fpn_tail": [_fp] * 6 + [_i] * 6 + [_s],
    "pmn_fpn_level": [_fp] * 6 + [_i] * 6 + [_s],
    "pmn_conv2d_f16s": [_fp] * 4 + [_i] * 8 + [_s],
    "pmn_offset_heads_f16s": [_fp] * 5 + [_i] * 7 + [_s],
    "pmn_conv2d_f16s_pair": [_fp] * 6 + [_i] * 5 + [_s],
    "pmn_refine_front": [_fp] * 7 + [_i] * 3 + [_s],
    "pmn_refine_tail": [_fp] * 8 + [_i] * 3 + [_s],
    "pmn_refine_fused": [_fp] * 13 + [_i] * 3 + [_s],
    "pmn_conv2d_mfma": [_fp] * 5 + [_i] * 12 + [_s],
    "pmn_deconv3x3s2": [_fp] * 4 + [_i] * 6 + [_s],
    "pmn_stage_projections": [_fp, _fp, _i, _i, _i, _f, _fp, _s],
    "pmn_stem": [_fp] * 6 + [_i] * 3 + [_s],
    "pmn_stem_f16s": [_fp] * 6 + [_i] * 3 + [_s],
    "pmn_stem_f16s_views": [_fp, _i] + [_fp] * 5 + [_i] * 3 + [_s],
    "pmn_differentiable_warping": [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp, _s],
    "pmn_fuse_view": [_fp, ctypes.c_longlong, _i, _hp, _hp, _i, _fp, _i, _i, _f, _f, _i, _f, _fp, _fp, _fp, _ip, _s],
    "pmn_pack_points": [_fp, _fp, _fp, _i, _i, _i, _fp, ctypes.c_longlong, _fp, _ip, _fp, _s],
    "pmn_normalize_depth": [_fp, _fp, _fp, _i, _i, _fp, _s],
    "pmn_check_f16_domain": [_fp, ctypes.c_longlong, _ip, _s],
    "pmn_plan_create": [ctypes.POINTER(ctypes.c_void_p)],
    "pmn_plan_begin": [_hp],
    "pmn_plan_end": [_hp],
    "pmn_plan_count": [_hp],
    "pmn_plan_kernel_name": [_hp, _i],
    "pmn_plan_launch": [_hp, _s],
    "pmn_plan_destroy": [_hp],
}

# libpmn_hip_experimental.so only (include/pmn_hip_experimental.h; `make -C patchmatchnet_amd/csrc EXPERIMENTAL=1`)
EXPERIMENTAL_SIGNATURES = {"pmn_set_tuning": [_i, _i],
                           "pmn_conv3x3_wino": [_fp] * 4 + [_i] * 5 + [_s],
                           "pmn_conv5x5s2_wino": [_fp] * 4 + [_i] * 6 + [_s]}
EXPERIMENTAL_LIB_PATH = os.path.join(_CSRC, "libpmn_hip_experimental.so")


def experimental() -> bool:
    """True when this process opted into the research build (PMN_EXPERIMENTAL=1): the product never sets it."""
    return os.environ.get("PMN_EXPERIMENTAL", "") == "1"


_LIB: Optional[ctypes.CDLL] = None


class PmnError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, "-j4"] + (["-B"] if force else []) + (["EXPERIMENTAL=1"] if experimental() else [])
    if not verbose:
        args.insert(1, "-s")
    subprocess.check_call(args)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Loads libpmn_hip.so; raises PmnError when it is missing (the product has no other compute path)."""
    global _LIB
    if _LIB is None:
        path, sigs = LIB_PATH, SIGNATURES
        if experimental():
            path, sigs = EXPERIMENTAL_LIB_PATH, {**SIGNATURES, **EXPERIMENTAL_SIGNATURES}
        if not os.path.isfile(path):
            raise PmnError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C patchmatchnet_amd/csrc` -- patchmatchnet_amd has no fallback path")
        # PyDLL: the entry points only enqueue kernels (microseconds, never a synchronisation), so they are called WITHOUT
        # releasing the GIL.  With CDLL each of the ~55 launches of a forward is a release/re-acquire, and every one of them is
        # an opening for eval.py's writer threads to take the GIL away from the launch thread (PMN_CTYPES=cdll restores that).
        L = (ctypes.CDLL if os.environ.get("PMN_CTYPES", "") == "cdll" else ctypes.PyDLL)(path)
        for name, argtypes in sigs.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise PmnError(f"libpmn_hip.so does not export {name}") from e
            fn.argtypes = argtypes
            fn.restype = ctypes.c_char_p if name in ("pmn_error_string", "pmn_plan_kernel_name") else ctypes.c_int
        if L.pmn_abi_version() != ABI_VERSION:
            raise PmnError(f"libpmn_hip.so ABI {L.pmn_abi_version()} != expected {ABI_VERSION}: rebuild")
        _LIB = L
    return _LIB


def check(code: int, what: str) -> None:
    if code != 0:
        raise PmnError(f"{what} failed: {lib().pmn_error_string(code).decode()} (code {code})")
