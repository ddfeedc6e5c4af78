"""ctypes binding of libw2e.so (include/w2e.h).  This is the ONLY door to compute for the hot path:
if the library is missing or a tensor is not on an MI355X the ops raise -- there is no CPU or
eager-PyTorch fallback (the CPU restatement lives in oracle/ and is test-only)."""
import ctypes
import os

import torch

from . import build as _build

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float

class DemodLayer(ctypes.Structure):  # w2e_demod_layer (include/w2e.h)
    _fields_ = [("s", ctypes.c_void_p), ("wsq", ctypes.c_void_p), ("d", ctypes.c_void_p), ("cin", ctypes.c_int), ("cout", ctypes.c_int)]


_PROTOS = {
    "w2e_version": (_I, []),
    "w2e_last_error": (ctypes.c_char_p, []),
    "w2e_set_option": (_I, [ctypes.c_char_p, ctypes.c_char_p]),
    "w2e_get_option": (_I, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]),
    "w2e_upfirdn2d": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _F, _F, _P]),
    "w2e_blur_adjoint_actbwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _F, _P]),
    "w2e_mapper_pixelnorm": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "w2e_mapper_linear": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _F, _F, _I, _P]),
    "w2e_mapper_wgrad": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _F, _F, _I, _P]),
    "w2e_mapper_gather": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "w2e_mapper_transpose": (_I, [_P, _I, _P, _P]),
    "w2e_ssmapper_pixelnorm": (_I, [_P, _P, _I, _I, _P, _P]),
    "w2e_ssmapper_gather": (_I, [_P, _P, _I, _I, _P, _P]),
    "w2e_ssmapper_linear": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _F, _P]),
    "w2e_ssmapper_wgrad": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _F, _P]),
    "w2e_bias_act_fwd": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _F, _F, _P]),
    "w2e_bias_act_bwd": (_I, [_P, _P, _P, _L, _F, _F, _P]),
    "w2e_bias_act_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _F, _F, _P]),
    "w2e_conv_pack": (_I, [_P, _P, _I, _I, _F, _I, _I, _P]),
    "w2e_modconv3x3": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "w2e_wino_weights_fused": (_I, [_P, _P, _I, _I, _P]),
    "w2e_wino_fused": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "w2e_wino_gemm_plan": (_I, [_I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64)]),
    "w2e_wino_pack_input": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "w2e_wino_gemm": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "w2e_demod_fwd": (_I, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "w2e_demod_all_fwd": (_I, [ctypes.POINTER(DemodLayer), _I, _I, _F, _P]),
    "w2e_demod_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "w2e_style_affine_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "w2e_style_affine_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "w2e_torgb_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "w2e_torgb_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "w2e_torgb_bwd_acc": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "w2e_torgb_styled_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "w2e_torgb_styled_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "w2e_torgb_bwd_actbwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "w2e_clip_preproc_fwd": (_I, [_P, _P, _L, _I, _P]),
    "w2e_clip_preproc_bwd": (_I, [_P, _P, _L, _I, _P]),
    "w2e_id_preproc_fwd": (_I, [_P, _P, _L, _I, _P]),
    "w2e_id_preproc_bwd": (_I, [_P, _P, _L, _I, _P]),
    "w2e_mask_blend_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "w2e_mask_blend_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
}

_lib = None


def header_version():
    """W2E_VERSION as include/w2e.h states it: the one place the ABI version is written down (the loader, the driver's
    build hook and the tests all compare the built library against THIS, so none of them can go stale on its own)."""
    import re
    h = os.path.join(os.path.dirname(_build.PKG), "include", "w2e.h")
    m = re.search(r"^#define\s+W2E_VERSION\s+(\d+)", open(h).read(), re.M)
    if not m:
        raise RuntimeError(f"{h}: no W2E_VERSION")
    return int(m.group(1))


def lib_path():
    return _build.LIB_PATH


def load():
    """Load libw2e.so (built in-tree by where2edit_amd.build).  Raises if it does not exist."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: run `python -m where2edit_amd.build` (hipcc, gfx950). "
                               "where2edit_amd has no fallback path without its HIP library.")
        lib = ctypes.CDLL(path)  # torch is already imported: libamdhip64.so.7 resolves to the runtime torch uses
        for name, (res, args) in _PROTOS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        from . import _lib_vit  # noqa: F401  (registers the ViT entry points when present)
        _lib_vit.declare(lib)
        from . import run_attention as _ra  # the region-attention entry points (include/w2e_attention.h)
        _ra.declare(lib)
        from . import irse_hip as _ir  # the IR-SE50 entry points (include/w2e_irse.h)
        _ir.declare(lib)
        if lib.w2e_version() != header_version():
            raise RuntimeError(f"libw2e.so is ABI {lib.w2e_version()}, include/w2e.h says {header_version()}: rebuild with `python -m where2edit_amd.build --force`")
        _lib = lib
    return _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_cur_dev = torch.cuda.current_device  # (lazy CUDA init happens long before the first kernel call)


def stream_ptr():
    """The current HIP stream of the current device as a void*.  (torch.cuda.current_stream() builds a Python Stream
    object per call, ~10 us -- 2 ms of host time per mapper step; the raw accessor is ~0.3 us.)"""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(_cur_dev()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous fp32 CUDA(HIP) tensor, or NULL for None.  The tensor must live on the CURRENT
    device: the kernels are launched on the current device's stream, and a pointer of another GPU there is a memory
    fault, not a Python error (wrap multi-device use in `with torch.cuda.device(t.device):`)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("where2edit_amd ops run on the GPU only (got a CPU tensor); the CPU restatement "
                           "is oracle/, for tests")
    if t.device.index != _cur_dev():
        raise RuntimeError(f"where2edit_amd: tensor on cuda:{t.device.index} but the current device is cuda:{_cur_dev()} "
                           "(kernels launch on the current device's stream): call torch.cuda.set_device(...) or use "
                           "`with torch.cuda.device(t.device):`")
    if t.dtype != torch.float32:
        raise RuntimeError(f"where2edit_amd kernels are fp32 (got {t.dtype})")
    if not t.is_contiguous():
        raise RuntimeError("internal: non-contiguous tensor passed to a kernel")
    # an empty tensor has no storage: hand the ABI a non-NULL dummy address (it is never dereferenced -- every entry
    # point returns before launching when a dimension is 0) so that "NULL" keeps meaning "argument absent"
    return ctypes.c_void_p(t.data_ptr() if t.numel() else 16)


def call(name, *args):
    """Invoke an entry point; a non-zero status becomes RuntimeError(w2e_last_error())."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.w2e_last_error().decode()}")


def set_option(name, value):
    """w2e_set_option: process-wide library options ("conv_precision", "deterministic", "tune_cfg", ...; include/w2e.h)."""
    lib = load()
    v = None if value is None else str(value).encode()
    if lib.w2e_set_option(name.encode(), v) != 0:
        raise RuntimeError(lib.w2e_last_error().decode())


def get_option(name):
    lib = load()
    out = ctypes.c_int(0)
    if lib.w2e_get_option(name.encode(), ctypes.byref(out)) != 0:
        raise RuntimeError(lib.w2e_last_error().decode())
    return out.value
