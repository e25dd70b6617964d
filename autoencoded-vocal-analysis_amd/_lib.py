"""ctypes binding of ``libava_hip.so`` (C ABI declared in ``include/ava_hip.h``).

The library is the only compute backend of this package.  There is no CPU or
PyTorch fallback: if the shared object is missing or a call returns an error the
caller gets an exception.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# AVA_HIP_LIB_TAG=x loads csrc/libava_hip_x.so instead: same-box A/B of two builds (tools/ab_prof.sh), nothing else
_TAG = os.environ.get("AVA_HIP_LIB_TAG", "")
LIB_PATH = os.path.join(CSRC, "libava_hip_%s.so" % _TAG if _TAG else "libava_hip.so")

_ERR = {-1: "AVA_EINVAL (bad argument / unsupported shape)", -2: "AVA_ELAUNCH (HIP launch failure)",
        -3: "AVA_EWORKSPACE (workspace too small)"}


class AvaHipError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise AvaHipError("building libava_hip.so failed")
    return LIB_PATH


_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_f = C.c_float
_d = C.c_double
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/ava_hip.h one to one
SIGNATURES = {
    "ava_version": (_i, []),
    "ava_arena_floats": (_i64, [_i]),
    "ava_param_offset": (_i64, [_i, _i, C.POINTER(_i64)]),
    "ava_workspace_bytes": (_sz, [_i, _i]),
    "ava_model_create": (_i, [C.POINTER(_p), _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _sz]),
    "ava_model_destroy": (None, [_p]),
    "ava_arena_floats_hw": (_i64, [_i, _i, _i]),
    "ava_param_offset_hw": (_i64, [_i, _i, _i, _i, C.POINTER(_i64)]),
    "ava_workspace_bytes_hw": (_sz, [_i, _i, _i, _i]),
    "ava_model_create_hw": (_i, [C.POINTER(_p), _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _sz]),
    "ava_model_create_ex": (_i, [C.POINTER(_p), _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _sz]),
    "ava_forward": (_i, [_p, _p, _i, _p, _p, _i, _p, _p, _p, _p]),
    "ava_forward_noise": (_i, [_p, _p, _i, _p, C.c_uint64, C.c_uint64, _i, _p, _p, _p, _p]),
    "ava_backward": (_i, [_p, _p, _i, _p]),
    "ava_set_backward_scale": (_i, [_p, _p]),
    "ava_backward_num_parts": (_i, []),
    "ava_backward_part": (_i, [_p, _p, _i, _i, _p]),
    "ava_grad_bucket": (_i, [_p, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "ava_adam_step": (_i, [_p, _d, _d, _d, _d, _i, _p]),
    "ava_adam_step_range": (_i, [_p, _i64, _i64, _d, _d, _d, _d, _i, _p]),
    "ava_encode": (_i, [_p, _p, _i, _i, _p, _p, _p, _p]),
    "ava_decode": (_i, [_p, _p, _i, _i, _p, _p]),
    "ava_last_z": (_p, [_p]),
    "ava_last_xrec": (_p, [_p]),
    "ava_debug_buffer": (_p, [_p, C.c_char_p, C.POINTER(_i64)]),
    "ava_debug_materialize": (_i, [_p, _p, _i, _p]),
    "ava_set_cu_reserve": (_i, [_i]),
    "ava_get_cu_reserve": (_i, []),
    "ava_model_set_cu_reserve": (_i, [_p, _i]),
    "ava_model_get_cu_reserve": (_i, [_p]),
    "ava_occupy_cus": (_i, [_i, _i, _f, _p]),
    "ava_profile_enable": (_i, [_p, _i]),
    "ava_profile_read": (_i, [_p, C.POINTER(_f), C.POINTER(_i)]),
    "ava_fill_normal": (_i, [_p, _i64, C.c_uint64, C.c_uint64, _p]),
    "ava_pack_conv_weight": (_i, [_p, _p, _i, _i, _i, _p]),
    "ava_conv_grid": (_i, [_i, _i, _i, _i]),
    "ava_conv3x3": (_i, [_p] * 13 + [_i] * 9 + [_f, _p]),
    "ava_conv3x3_wgrad": (_i, [_p] * 9 + [_i] * 7 + [_p]),
    "ava_conv_wgrad_grid": (_i, [_i, _i, _i, _i]),
    "ava_conv_wgrad_rows": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "ava_conv_fused_grid": (_i, [_i, _i, _i, _i, _i, _i]),
    "ava_conv3x3_bwd_fused": (_i, [_p] * 14 + [_i] * 7 + [_p]),
    "ava_conv_wgrad_reduce": (_i, [_p, _i, _p, _p, _i, _i, _i, _p]),
    "ava_bn_stats": (_i, [_p, _i64, _i, _p, C.POINTER(_i), _p]),
    "ava_bn_finalize": (_i, [_p, _i, _i64, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "ava_bn_finalize_bwd": (_i, [_p, _i, _i64, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "ava_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "ava_gemm": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "ava_latent_fwd": (_i, [_p] * 9 + [_i, _i, _p]),
    "ava_latent_bwd": (_i, [_p] * 9 + [_i, _i, _p]),
    "ava_elbo_finalize": (_i, [_p, _i, _p, _i, _i, _f, _p, _p]),
    "ava_adam_flat": (_i, [_p, _p, _p, _p, _i64, _d, _d, _d, _d, _i, _p]),
    "ava_cast_to_f32": (_i, [_p, _i, _i64, _p, _p]),
    "ava_host_gather_rows": (_i, [_p, _p, _p, _i64, _i64, _sz, _i]),
    "ava_mmd2_workspace_bytes": (_sz, [_i, _i]),
    "ava_mmd2": (_i, [_p, _i, _p, _i, _p, _i, _d, _p, _p, _sz, _p]),
    "ava_mmd2_linear": (_i, [_p, _i, _p, _p, _i, _d, _p, _p, _sz, _p]),
    "ava_pair_sqdist": (_i, [_p, _i, _p, _p, _i, _p, _p]),
    "ava_spec_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "ava_get_spec_batch": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _d, _i, _i, _p, _d, _p, _i, _i, _d, _d, _d, _i,
                                _i, _i, _d, _p, _p, _p, _sz, _p]),
}

_lib = None


def load():
    """Load libava_hip.so (after torch, so that both share the HIP runtime torch ships)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AvaHipError(
            "%s not found: the HIP extension is the only backend of this package. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C %s`." % (LIB_PATH, CSRC))
    import torch  # noqa: F401  (loads torch's libamdhip64.so first; same SONAME as ROCm's)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise AvaHipError("%s failed: %s" % (what, _ERR.get(rc, rc)))


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
