"""Generates tests/golden/attention_net.npz by running the REFERENCE's region-attention mapper net
(attention/run_attention.py:703-893, FullSpaceMapperFEATClusterLinStyle_Net) in this container.

    python tests/golden/make_golden_attention.py        (needs /root/reference; never runs on the GPU box)

attention/run_attention.py imports packages that are absent from this image (torchvision, clip, tensorboard,
torch_fidelity, tqdm is present).  As in SURVEY 8(c), empty stub modules stand in for them in sys.modules -- none of
their code is on the path of the net's forward EXCEPT torchvision.transforms.functional.gaussian_blur (run_attention.py:884).
That one call is served by a harness-side function implementing torchvision's published algorithm (kernel_size 5 ->
sigma = 0.3*((5-1)*0.5-1)+0.8 = 1.1, separable gaussian, reflect padding); the harness also RECORDS the tensor the
reference hands to it, so everything up to and including the straight-through threshold is pinned by the reference
itself ("pre_blur") and only the blur is pinned by the published definition.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
import seeded  # noqa: E402

RECORD = {}


def gaussian_blur5(img, kernel_size):
    """torchvision.transforms.functional.gaussian_blur(img, 5): sigma 1.1, reflect padding, depthwise separable."""
    assert kernel_size == 5
    RECORD["pre_blur"] = img.detach().clone()
    sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, steps=kernel_size)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :]).to(img.dtype)
    c = img.shape[-3]
    pad = kernel_size // 2
    x4 = torch.nn.functional.pad(img, [pad, pad, pad, pad], mode="reflect")
    return torch.nn.functional.conv2d(x4, k2.expand(c, 1, kernel_size, kernel_size), groups=c)


def import_reference_net():
    torch.Tensor.cuda = lambda self, *a, **k: self
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
                 "torchvision.models", "clip", "torch_fidelity", "torch.utils.tensorboard", "sklearn.metrics"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    tv.transforms = sys.modules["torchvision.transforms"]
    tv.transforms.functional = sys.modules["torchvision.transforms.functional"]
    tv.transforms.functional.gaussian_blur = gaussian_blur5
    tv.utils = sys.modules["torchvision.utils"]
    tv.utils.save_image = None
    tv.models = sys.modules["torchvision.models"]
    sys.modules["torch_fidelity"].calculate_metrics = None
    sys.modules["torch.utils.tensorboard"].SummaryWriter = None
    sys.modules["sklearn.metrics"].jaccard_score = None
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "attention"))
    import run_attention
    return run_attention


# ---- the seeded problem (regenerated bit-exactly by the tests from these names) ---------------------------------------
LAYERS = 14           # n_latent of a 256^2 generator: 20 S-space codes, 14 feature groups (the net's dims are size-agnostic)
ATT_LAYER = 7         # style_layers[7] = 8 -> codes 0..7 are edited
CLUSTER_LAYER = 7     # features of layer 7 (16x16 at 256^2) are clustered
CLUSTERS = 6
SIZE = 16             # blend_size = resolution of layer ATT_LAYER
BATCH = 2
DIMS = [512] * 15 + [256] * 3 + [128] * 3
FEAT_RES = [4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64, 128, 128, 128, 256, 256, 256]  # 20 layers of G(256)
FEAT_CH = [512] * 14 + [256] * 3 + [128] * 3


def feature_maps():
    """20 post-layer activations of a 256^2 generator (shapes only; seeded values) + the const input appended by the caller
    (run_attention.py:1110).  Layer CLUSTER_LAYER-1 gets a block structure so that the nearest-centroid assignment is
    well separated (each pixel = one of CLUSTERS prototypes + noise)."""
    feats = []
    for i, (r, c) in enumerate(zip(FEAT_RES, FEAT_CH)):
        ch = 3 if i in (1, 4, 7, 10, 13, 16, 19) else c  # ToRGB outputs
        feats.append(seeded.tensor(f"att.feat{i}", (BATCH, ch, r, r)))
    protos = seeded.tensor("att.protos", (CLUSTERS, 512), 1.0)
    labels = cluster_labels()
    f = protos[labels].permute(0, 3, 1, 2) + 0.25 * seeded.tensor("att.feat_noise", (BATCH, 512, SIZE, SIZE))
    feats[CLUSTER_LAYER - 1] = f.contiguous()
    feats.append(seeded.tensor("att.const", (1, 512, 4, 4)).repeat(BATCH, 1, 1, 1))
    return feats


def cluster_labels():
    yy, xx = torch.meshgrid(torch.arange(SIZE), torch.arange(SIZE), indexing="ij")
    lab0 = ((yy // 6) * 3 + (xx // 6)) % CLUSTERS
    lab1 = ((yy // 4) + (xx // 8) * 2) % (CLUSTERS - 1)  # sample 1 leaves cluster 5 EMPTY (NaN-mean branch, :859)
    return torch.stack([lab0, lab1])


def centroids():
    protos = seeded.tensor("att.protos", (CLUSTERS, 512), 1.0)
    pos = 0.05 * seeded.tensor("att.centroid_pos", (CLUSTERS, 64))
    return torch.cat([protos, pos], 1)


def net_state_dict(net):
    sd = {}
    for k, v in net.state_dict().items():
        if k == "initial_state":
            sd[k] = centroids()
        elif k == "initial_bias":
            sd[k] = torch.tensor([1.35])
        elif k.endswith("noise.weight"):
            sd[k] = torch.zeros_like(v)  # NoiseInjection draws fresh randn when noise=None: keep its strength 0 (the init)
        elif v.ndim >= 2:
            sd[k] = seeded.tensor("att.sd." + k, v.shape, 1.0)
        elif k.endswith(".bias") and ("modulation" in k or k.startswith("attention_textca") or
                                      (k.startswith("mapper_") and "text" not in k)):
            sd[k] = seeded.tensor("att.sd." + k, v.shape, 0.1, 1.0)
        else:
            sd[k] = seeded.tensor("att.sd." + k, v.shape, 0.1)
    return sd


def inputs():
    text = seeded.tensor("att.text", (BATCH, 512), 0.3)
    att_text = seeded.tensor("att.att_text", (1, 512), 0.3).repeat(BATCH, 1)
    n_codes = LAYERS + (LAYERS - 2) // 2
    styles = [seeded.tensor(f"att.style{c}", (BATCH, 1, DIMS[c]), 0.5, 1.0) for c in range(n_codes)]
    x = [torch.cat([text.unsqueeze(1), s], -1) for s in styles]
    return x, att_text, styles


def main():
    ra = import_reference_net()
    torch.manual_seed(0)
    net = ra.FullSpaceMapperFEATClusterLinStyle_Net(LAYERS, 1024, 512, attention_layer=ATT_LAYER, channel_multiplier=2,
                                                    cluster_layer=CLUSTER_LAYER, clusters=CLUSTERS, cluster_dim=576)
    sd = net_state_dict(net)
    net.load_state_dict(sd, strict=True)
    net.train()
    x, att_text, styles = inputs()
    feats = feature_maps()
    out, final_map, losses = net(x, feats, SIZE, attention_text=att_text)
    # gradients of a scalar of the new styles + the losses into a few mapper parameters
    r = [seeded.tensor(f"att.r{c}", tuple(o.shape)) for c, o in enumerate(out)]
    scalar = sum((o * rr).sum() for o, rr in zip(out, r)) + 3.0 * losses[0]
    names = ["mapper_0.weight", "mapper_0.bias", "mapper_text_3.0.weight", "mapper_text_3.1.bias", "mapper_all_7.weight", "mapper_all_7.bias"]
    params = dict(net.named_parameters())
    grads = torch.autograd.grad(scalar, [params[n] for n in names])
    store = {"final_map": final_map.detach().numpy(), "pre_blur": RECORD["pre_blur"].numpy(),
             "loss_delta": np.float64(losses[0].item()), "loss_reg": np.float64(losses[1].item()), "loss_tv": np.float64(losses[2].item()),
             "keys": np.asarray(sorted(sd)), "shapes": np.asarray([str(tuple(sd[k].shape)) for k in sorted(sd)]),
             "grad_names": np.asarray(names)}
    for c, o in enumerate(out):
        store[f"out{c}"] = o.detach().numpy()
    for n, g in zip(names, grads):
        store["grad." + n] = g.numpy()
    # the cluster assignment the reference computed (re-derived with its own pairwise_distance, utils.py:244-263)
    from utils import pairwise_distance
    bf = feats[CLUSTER_LAYER - 1]
    cs = bf.shape[2]
    xs = torch.arange(cs).float().unsqueeze(0).repeat(cs, 1) * 2 / float(cs - 1) - 1
    ys = torch.arange(cs).float().unsqueeze(1).repeat(1, cs) * 2 / float(cs - 1) - 1
    cat = torch.cat([bf, xs[None, None].repeat(BATCH, 32, 1, 1), ys[None, None].repeat(BATCH, 32, 1, 1)], 1)
    dis = pairwise_distance(cat.permute(0, 2, 3, 1).reshape(-1, 576), sd["initial_state"])
    store["assign"] = torch.argmin(dis, 1).view(BATCH, cs, cs).numpy().astype(np.int32)
    store["assign_dis_sample"] = dis[:64].numpy()
    np.savez_compressed(os.path.join(HERE, "attention_net.npz"), **store)
    print("saved attention_net.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(store.items())[:8]})
    print("losses", [l.item() for l in losses], "map range", final_map.min().item(), final_map.max().item())
    print("assign counts", [np.bincount(store["assign"][b].ravel(), minlength=CLUSTERS).tolist() for b in range(BATCH)])


if __name__ == "__main__":
    main()
