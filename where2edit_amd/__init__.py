"""where2edit_amd -- MI355X-native hot path of Where2edit (StyleGAN2 generator stack, region-attention
blend, CLIP ViT-B/32 image encoder) behind the reference's Python module/operator surface.

Layout:
  csrc/ + lib/libw2e.so   hand-written HIP (gfx950) behind the C ABI of include/w2e.h
  _lib.py                 ctypes door to that library (no fallback)
  functional.py           torch.autograd.Functions over the C ABI
  op/                     models/stylegan2/op seam: FusedLeakyReLU, fused_leaky_relu, upfirdn2d
  stylegan2.py            models/stylegan2/model.py surface (Generator, ModulatedConv2d, ...)
  attention_model.py      attention/attention_model.py Generator (features + region blend)
  latent_mappers.py, styleclip_mapper.py   mapper/ surface
  clip_vit.py, clip_loss.py                criteria/clip_loss.py surface + ViT-B/32
  coach.py, ranger.py     the mapper training step (mapper/training/coach.py:70-92) + optimizer
  dist.py                 data-parallel step: shard latents, one RCCL all-reduce of mapper grads
"""
__version__ = "0.1.0"


def set_deterministic(enabled=True):
    """Bit-reproducible results, the counterpart of the reference's `cudnn.deterministic = True`
    (attention/run_attention.py:903-904): libw2e.so stops using fp32 atomics (no split-K, fixed-order reductions;
    w2e_set_option("deterministic")) and PyTorch's own ops are switched to their deterministic algorithms (rocBLAS
    without atomics).  Costs a few percent of throughput; off by default."""
    import torch
    from . import _lib
    _lib.set_option("deterministic", 1 if enabled else 0)
    torch.use_deterministic_algorithms(bool(enabled), warn_only=True)
