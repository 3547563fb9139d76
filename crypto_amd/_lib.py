"""ctypes binding of libbn256msm.so (include/bn256msm.h).  Fails loudly if the HIP library
is missing: there is no CPU fallback on the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

# Every lane (call in flight) has a HIP stream of its own, and a HIP process serves all of its streams from
# GPU_MAX_HW_QUEUES hardware queues (4 by default): two lanes on one queue run one after the other
# (tools/probe/stream_probe.hip).  The runtime reads the variable when it initialises, so it
# is raised here, at import, before this package makes the first HIP call -- WNLA 2^20 22.8 -> 19.7 ms, go/bp
# N = 64 prove 7.5 -> 7.1 ms, four 2^20-point calls in flight +0.6...2 %.  CRYPTO_AMD_HW_QUEUES=0 leaves the
# environment alone; a C or Go caller sets the variable itself.
# SIDE EFFECT: the variable is process-wide, so torch's and RCCL's streams in the same process are served from
# the raised number of queues too.  It has NO effect when the HIP runtime was initialised before this import
# (a caller that ran torch.cuda first): that case is detected below and reported with a warning, because the
# figures quoted in DESIGN.md for several calls in flight depend on the raised value.
_want_queues = int(os.environ.get("CRYPTO_AMD_HW_QUEUES", "8"))
if _want_queues and int(os.environ.get("GPU_MAX_HW_QUEUES") or 4) < _want_queues:
    _hip_up = False
    try:
        import sys as _sys
        _t = _sys.modules.get("torch")
        _hip_up = bool(_t is not None and _t.cuda.is_initialized())
    except Exception:  # pragma: no cover
        _hip_up = False
    if _hip_up:
        import warnings
        warnings.warn(
            f"crypto_amd: the HIP runtime was initialised before this import, so GPU_MAX_HW_QUEUES={_want_queues} "
            "cannot take effect in this process (lanes will share hardware queues and overlap less); import "
            "crypto_amd before the first torch.cuda call, or export GPU_MAX_HW_QUEUES yourself", RuntimeWarning)
    else:
        os.environ["GPU_MAX_HW_QUEUES"] = str(_want_queues)
LIB_PATH = os.environ.get("BN256MSM_LIB") or os.path.join(_HERE, "libbn256msm.so")  # override: A/B builds

EXPORTS = [
    "bn256_ctx_create", "bn256_ctx_create_multi", "bn256_ctx_destroy", "bn256_ctx_stream", "bn256_bases_upload", "bn256_bases_free",
    "bn256_bases_len", "bn256_msm", "bn256_msm_dev", "bn256_msm_dev_noalloc", "bn256_msm_submit", "bn256_msm_wait", "bn256_msm_multi", "bn256_msm_multi_strided", "bn256_fold2", "bn256_scalar_mul_batch",
    "bn256_add_batch", "bn256_fold", "bn256_bases_download", "bn256_random_bases", "bn256_random_scalars",
    "bn256_g1_sum", "bn256_ctx_set_tuning", "bn256_ctx_set_phase_timing", "bn256_last_timing", "bn256_selftest", "bn256_strerror",
    "bn256_fr_upload", "bn256_fr_random", "bn256_fr_download", "bn256_fr_len", "bn256_fr_device_ptr", "bn256_fr_free",
    "bn256_fr_fold", "bn256_fr_dot", "bn256_fr_tensor_expand_multi", "bn256_fr_tensor_step_multi", "bn256_ctx_fr_defer_sync", "bn256_fr_fold_multi", "bn256_fr_dot_multi", "bn256_fr_dot_multi_dev", "bn256_msm_fr", "bn256_msm_fr_multi", "bn256_com_batch", "bn256_glv_split", "bn256_msm_fr_multi_submit", "bn256_msm_multi_submit", "bn256_fold2_submit", "bn256_fold_wait", "bn256_ctx_drain",
]

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m crypto_amd.build` "
            "(hipcc --offload-arch=gfx950); crypto_amd has no CPU fallback")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  Two HIP runtimes in one process
    # cannot both own the GPU ("No HIP GPUs are available" from whichever comes second), so when
    # torch is importable load it first: our NEEDED libamdhip64.so.7 then binds to the copy that
    # is already mapped.  Without torch the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less hosts
        pass
    lib = C.CDLL(LIB_PATH)
    vp, sz, u8p, i = C.c_void_p, C.c_size_t, C.c_char_p, C.c_int
    lib.bn256_ctx_create.argtypes = [i, C.POINTER(vp)]
    lib.bn256_ctx_create_multi.argtypes = [C.POINTER(C.c_int), i, C.POINTER(vp)]
    lib.bn256_ctx_destroy.argtypes = [vp]
    lib.bn256_ctx_destroy.restype = None
    lib.bn256_ctx_stream.argtypes = [vp]
    lib.bn256_ctx_stream.restype = vp
    lib.bn256_bases_upload.argtypes = [vp, u8p, sz, i, C.POINTER(vp)]
    lib.bn256_bases_free.argtypes = [vp, vp]
    lib.bn256_bases_free.restype = None
    lib.bn256_bases_len.argtypes = [vp]
    lib.bn256_bases_len.restype = sz
    lib.bn256_msm.argtypes = [vp, vp, sz, sz, u8p, u8p]
    lib.bn256_msm_dev.argtypes = [vp, vp, sz, sz, vp, u8p]
    lib.bn256_msm_dev_noalloc.argtypes = [vp, vp, sz, sz, vp, u8p]
    lib.bn256_msm_submit.argtypes = [vp, vp, sz, sz, vp, C.POINTER(C.c_int)]
    lib.bn256_msm_wait.argtypes = [vp, C.c_int, u8p]
    lib.bn256_msm_multi.argtypes = [vp, i, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), C.POINTER(u8p), u8p]
    lib.bn256_msm_multi_strided.argtypes = [vp, i, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz),
                                            C.POINTER(u8p), u8p]
    lib.bn256_fold2.argtypes = [vp, vp, sz, sz, vp, sz, sz, sz, u8p, u8p, C.POINTER(vp)]
    lib.bn256_scalar_mul_batch.argtypes = [vp, vp, sz, sz, u8p, i, C.POINTER(vp)]
    lib.bn256_add_batch.argtypes = [vp, vp, sz, vp, sz, sz, C.POINTER(vp)]
    lib.bn256_fold.argtypes = [vp, vp, sz, vp, sz, sz, sz, u8p, u8p, C.POINTER(vp)]
    lib.bn256_bases_download.argtypes = [vp, vp, sz, sz, u8p]
    lib.bn256_random_bases.argtypes = [vp, C.c_uint64, sz, sz, i, C.POINTER(vp)]
    lib.bn256_random_scalars.argtypes = [vp, C.c_uint64, sz, sz, vp]
    lib.bn256_fr_upload.argtypes = [vp, u8p, sz, C.POINTER(vp)]
    lib.bn256_fr_random.argtypes = [vp, C.c_uint64, sz, sz, C.POINTER(vp)]
    lib.bn256_fr_download.argtypes = [vp, vp, sz, sz, u8p]
    lib.bn256_fr_len.argtypes = [vp]
    lib.bn256_fr_len.restype = sz
    lib.bn256_fr_device_ptr.argtypes = [vp]
    lib.bn256_fr_device_ptr.restype = vp
    lib.bn256_fr_free.argtypes = [vp, vp]
    lib.bn256_fr_free.restype = None
    lib.bn256_fr_fold.argtypes = [vp, vp, sz, sz, vp, sz, sz, sz, u8p, u8p, C.POINTER(vp)]
    lib.bn256_fr_dot.argtypes = [vp, vp, sz, sz, sz, vp, sz, sz, sz, u8p, u8p]
    psz, pvp = C.POINTER(sz), C.POINTER(vp)
    lib.bn256_fr_fold_multi.argtypes = [vp, i, pvp, psz, psz, pvp, psz, psz, psz, u8p, u8p, pvp]
    lib.bn256_fr_dot_multi.argtypes = [vp, i, pvp, psz, psz, psz, pvp, psz, psz, psz, C.POINTER(u8p), u8p]
    if hasattr(lib, "bn256_fr_dot_multi_dev"):  # absent from an older build loaded through BN256MSM_LIB (A/B runs)
        lib.bn256_fr_dot_multi_dev.argtypes = [vp, i, pvp, psz, psz, psz, pvp, psz, psz, psz, C.POINTER(u8p), pvp]
    if hasattr(lib, "bn256_fr_tensor_step_multi"):  # absent from an older build loaded through BN256MSM_LIB (A/B runs)
        lib.bn256_ctx_fr_defer_sync.argtypes = [vp, i]
        lib.bn256_fr_tensor_step_multi.argtypes = [vp, i, pvp, i, u8p, u8p, pvp]
        lib.bn256_fr_tensor_expand_multi.argtypes = [vp, i, pvp, psz, pvp, i, C.POINTER(C.c_int), C.POINTER(C.c_int), u8p, u8p, pvp]
    lib.bn256_msm_fr.argtypes = [vp, vp, sz, sz, sz, vp, sz, sz, u8p]
    lib.bn256_msm_fr_multi.argtypes = [vp, i, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(vp),
                                       C.POINTER(sz), C.POINTER(sz), C.POINTER(u8p), u8p]
    lib.bn256_com_batch.argtypes = [vp, u8p, u8p, u8p, u8p, sz, C.POINTER(vp)]
    lib.bn256_msm_fr_multi_submit.argtypes = [vp, i, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz),
                                              C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), C.POINTER(u8p),
                                              C.POINTER(C.c_int)]
    lib.bn256_msm_multi_submit.argtypes = [vp, i, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), C.POINTER(u8p),
                                           C.POINTER(C.c_int)]
    lib.bn256_fold2_submit.argtypes = [vp, vp, sz, sz, vp, sz, sz, sz, u8p, u8p, C.POINTER(vp), C.POINTER(C.c_int)]
    lib.bn256_fold_wait.argtypes = [vp, i]
    lib.bn256_ctx_drain.argtypes = [vp]
    lib.bn256_glv_split.argtypes = [u8p, u8p, C.POINTER(C.c_int), u8p, C.POINTER(C.c_int)]
    lib.bn256_g1_sum.argtypes = [u8p, sz, u8p]
    lib.bn256_ctx_set_tuning.argtypes = [vp, i, i]
    lib.bn256_ctx_set_phase_timing.argtypes = [vp, i]
    lib.bn256_last_timing.argtypes = [vp, C.POINTER(C.c_float), i]
    lib.bn256_selftest.argtypes = [vp]
    lib.bn256_strerror.argtypes = [i]
    lib.bn256_strerror.restype = C.c_char_p
    _lib = lib
    return lib
