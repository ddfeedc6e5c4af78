"""ctypes prototypes of the CLIP-ViT entry points of libw2e.so (include/w2e_vit.h)."""
import ctypes

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

PROTOS = {}


def declare(lib):
    for name, (res, args) in PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
