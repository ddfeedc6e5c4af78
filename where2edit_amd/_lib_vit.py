"""ctypes prototypes of the CLIP-ViT entry points of libw2e.so (include/w2e_vit.h)."""
import ctypes

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

PROTOS = {
    "w2e_gemm": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "w2e_gemm_ex": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "w2e_layernorm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _F, _P]),
    "w2e_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "w2e_layernorm_bwd_add": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "w2e_attn_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "w2e_attn_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "w2e_reduce_gelu": (_I, [_P, _I, _L, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "w2e_reduce_ln_fwd": (_I, [_P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P]),
    "w2e_layernorm_bwd_part": (_I, [_P, _I, _L, _P, _P, _P, _P, _P, _P, _L, _I, _P, _I, _P]),
    "w2e_attn2_fwd": (_I, [_P, _I, _L, _P, _P, _I, _I, _I, _I, _P]),
    "w2e_attn2_bwd": (_I, [_P, _I, _L, _P, _P, _I, _L, _P, _I, _I, _I, _I, _P]),
    "w2e_pack_kq": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "w2e_gemm_pk": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "w2e_gemm_pk_splits": (_I, [_I, _I, _I]),
    "w2e_pack_kq_h": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "w2e_gemm_pk_h": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "w2e_gemm_pk_h_splits": (_I, [_I, _I, _I]),
    "w2e_clip_logits_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "w2e_clip_logits_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "w2e_step_loss_fwd": (_I, [_P, _I, _P, _P, _L, _F, _F, _P, _P]),
    "w2e_step_loss_bwd": (_I, [_P, _I, _P, _P, _L, _F, _F, _P, _P, _P]),
}


def declare(lib):
    for name, (res, args) in PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
