"""On-disk formats of the reference's artefacts, so existing Where2edit files load and what this package writes loads
there:

  * Coach checkpoints  {'state_dict': net.state_dict(), 'opts': vars(opts)}   (coach.py:163-172, 267-272;
    read back by mapper/scripts/inference.py:29-38 and StyleCLIPMapper.load_weights, styleclip_mapper.py:37-46)
  * region-attention mapper checkpoints: `torch.save(Mapper.state_dict(), ...)` of the DDP-wrapped net, i.e. every key
    carries a `module.` prefix (run_attention.py:1437, 1486); the demo strips it (show_demo/try_demo.py:38-42)
  * the k-means centres: `pickle.dump(torch.from_numpy(kmeans.cluster_centers_), f)` of a [K, 576] tensor
    (clustering_feature.py:394-397), read with `pickle.load` (run_attention.py:996-1003)
  * StyleGAN2 generator files {'g_ema': state_dict, ...} (run_attention.py:979-986)."""
import pickle

import torch


def strip_module_prefix(state_dict):
    """try_demo.py:38-42: 'module.xyz' -> 'xyz' (keys without the prefix pass through)."""
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def add_module_prefix(state_dict):
    return {(k if k.startswith("module.") else "module." + k): v for k, v in state_dict.items()}


def save_mapper(mapper, path, ddp_prefix=True):
    """What run_attention.py:1437 writes: the state_dict as DistributedDataParallel exposes it (`module.`-prefixed)."""
    sd = {k: v.detach().cpu() for k, v in mapper.state_dict().items()}
    torch.save(add_module_prefix(sd) if ddp_prefix else sd, path)


def load_mapper(mapper, path, strict=True):
    """Loads a mapper checkpoint written by the reference (DDP-prefixed) or by a single-process run (bare keys)."""
    sd = torch.load(path, map_location="cpu")
    return mapper.load_state_dict(strip_module_prefix(sd), strict=strict)


def save_clusters(centers, path):
    """clustering_feature.py:395-397: a pickled CPU tensor [K, C + 2*(C//16)] (float64 there: sklearn's dtype)."""
    with open(path, "wb") as f:
        pickle.dump(centers.detach().cpu(), f)


def load_clusters(path):
    """run_attention.py:996-1003"""
    with open(path, "rb") as f:
        centers = pickle.load(f)
    if not torch.is_tensor(centers) or centers.ndim != 2:
        raise ValueError(f"{path}: expected a pickled [K, D] tensor of k-means centres")
    return centers


def load_generator_weights(generator, path):
    """run_attention.py:982-986 / styleclip_mapper.py:44-46: ckpt['g_ema'], strict=False."""
    ckpt = torch.load(path, map_location="cpu")
    res = generator.load_state_dict(ckpt["g_ema"], strict=False)
    from .stylegan2 import freeze_conv_weights  # a loaded generator is a frozen decoder on this path (no conv-weight gradient
    freeze_conv_weights(generator)              # kernels: stylegan2._trainable_weight); every other parameter keeps its flag
    return res


def save_coach_checkpoint(net, opts, path):
    """coach.py:267-272"""
    torch.save({"state_dict": net.state_dict(), "opts": dict(vars(opts))}, path)


def load_coach_checkpoint(path):
    """inference.py:29-32: returns (state_dict, opts dict)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    return ckpt["state_dict"], ckpt["opts"]
