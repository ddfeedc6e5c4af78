"""Oracle (CPU, test-only): the region-attention mapper net of the paper's trainer,
`FullSpaceMapperFEATClusterLinStyle_Net` (attention/run_attention.py:703-893), functional over its state_dict,
plus the pieces it borrows: `pairwise_distance` (utils.py:244-263), the torchvision `gaussian_blur(x, 5)` it ends with
(kernel 5 -> sigma 1.1, reflect padding; torchvision is not in the image: published definition, see
tests/golden/make_golden_attention.py), `GatherLayer` + the InfoNCE term of the loop (utils.py:114-131,
run_attention.py:1312-1318).  Pinned by tests/golden/attention_net.npz, captured from the reference's own class."""
import math

import torch
import torch.nn.functional as F

from . import ops, stylegan2

LAYER_NUM = (0, 2, 3, 5, 6, 8, 9, 11, 12, 14, 15, 17, 18, 20, 21, 23, 24)  # run_attention.py:710: the conv layers
STYLE_LAYERS = (0, 2, 2, 3, 5, 5, 6, 8, 8, 9, 11, 11, 12, 14, 14, 15, 17, 17, 18, 20, 20, 21, 23, 23, 24, 26, 26)  # :711


def dims(channel_multiplier=2):
    cm = channel_multiplier
    return [512] * 12 + [256 * cm] * 3 + [128 * cm] * 3 + [64 * cm] * 3 + [32 * cm] * 3 + [16 * cm] * 3  # :709


def pairwise_distance(a, b):
    """utils.py:244-263: sum_m (a[n,m] - b[k,m])^2 through the [N,K,M] broadcast."""
    return ((a.unsqueeze(1) - b.unsqueeze(0)) ** 2.0).sum(-1)


def position_channels(batch, size, n):
    xs = torch.arange(size).float().unsqueeze(0).repeat(size, 1) * 2 / float(size - 1) - 1
    ys = torch.arange(size).float().unsqueeze(1).repeat(1, size) * 2 / float(size - 1) - 1
    return xs[None, None].repeat(batch, n, 1, 1), ys[None, None].repeat(batch, n, 1, 1)


def assign_clusters(blend_feature, initial_state):
    """run_attention.py:775-793: [B,C,s,s] features + C/16 x-position and C/16 y-position channels -> nearest centroid
    per pixel, [B,s,s] int64 in [0,K)."""
    b, c, s, _ = blend_feature.shape
    xp, yp = position_channels(b, s, c // 16)
    cat = torch.cat([blend_feature, xp, yp], 1).permute(0, 2, 3, 1).reshape(-1, c + 2 * (c // 16))
    return torch.argmin(pairwise_distance(cat, initial_state), 1).view(b, s, s)


def gaussian_blur5(x):
    k = 5
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    t = torch.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, steps=k)
    pdf = torch.exp(-0.5 * (t / sigma).pow(2))
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :]).to(x.dtype)
    c = x.shape[-3]
    return F.conv2d(F.pad(x, [2, 2, 2, 2], mode="reflect"), k2.expand(c, 1, k, k), groups=c)


def _styled_1x1(sd, pre, feat, style):
    """StyledConv(C, Cout, 1, C) called with input_is_stylespace=True and noise=None (:800, :830, :838); the noise strength
    is whatever the state_dict holds (0 at init; NoiseInjection then adds 0 * randn)."""
    out, _ = stylegan2.styled_conv(sd, pre, feat, style.view(feat.shape[0], 1, -1, 1, 1), torch.zeros(1, 1, feat.shape[2], feat.shape[3]),
                                   upsample=False, input_is_stylespace=True)
    return out


def forward(sd, x, feature_map, size, attention_text=None, *, attention_layer, cluster_layer, clusters, latent_dim=512,
            channel_multiplier=2):
    """FullSpaceMapperFEATClusterLinStyle_Net.forward (:754-893).  x: list of [B,1,latent_dim + dim_c] (text (+) style);
    feature_map: the generator's post-layer activations + the const input appended last (:1110).
    Returns (new styles [B,1,dim_c,1,1] list, final attention map [B,1,size,size], [loss_delta, loss_reg, loss_tv],
    extras dict with the intermediate maps)."""
    batch = x[0].shape[0]
    mapper_layer = STYLE_LAYERS[attention_layer]
    x_text = x[0][:, 0, :latent_dim]
    if attention_text is None:
        attention_text = x_text
    with torch.no_grad():
        choice = assign_clusters(feature_map[cluster_layer - 1], sd["initial_state"])
        choice = F.interpolate(choice.unsqueeze(1).float(), size).squeeze(1).long()  # nearest (:793)
    style = ops.equal_linear(attention_text, sd["attention_textca_first.weight"], sd["attention_textca_first.bias"])
    att = [F.interpolate(_styled_1x1(sd, "attention_first", feature_map[-1], style), size)]
    out = []
    loss_delta = 0
    for c in range(len(x)):
        x_c = x[c][:, :, latent_dim:]
        if c < mapper_layer:
            h = x_text
            for j in (0, 1):  # mapper_text_c: two EqualLinear(lr_mul=1, fused_lrelu) (:719-720)
                h = ops.equal_linear(h, sd[f"mapper_text_{c}.{j}.weight"], sd[f"mapper_text_{c}.{j}.bias"], activation=True)
            x_c_hidden = ops.equal_linear(x_c, sd[f"mapper_{c}.weight"], sd[f"mapper_{c}.bias"])
            mixed = ops.equal_linear(torch.cat([x_c_hidden, h.unsqueeze(1)], -1), sd[f"mapper_all_{c}.weight"], sd[f"mapper_all_{c}.bias"])
            x_c_new = x_c + 0.1 * (mixed - x_c)  # :820
            loss_delta = loss_delta + torch.mean(torch.norm(x_c_new - x_c, dim=-1)) / float(mapper_layer)
            out.append(x_c_new.unsqueeze(3).unsqueeze(3))
        else:
            out.append(x_c.unsqueeze(3).unsqueeze(3))
        if c in LAYER_NUM:
            style = ops.equal_linear(attention_text, sd[f"attention_textca_{c}.weight"], sd[f"attention_textca_{c}.bias"])
            att.append(F.interpolate(_styled_1x1(sd, f"attention_{c}", feature_map[c], style), size))
    each = torch.cat(att, 1)
    style = ops.equal_linear(attention_text, sd["attention_textca_last.weight"], sd["attention_textca_last.bias"])
    each = torch.sigmoid(_styled_1x1(sd, "attention_last", each, style) + sd["initial_bias"]).view(batch, size, size)
    # per-(sample, cluster) mean over the pixels assigned to it (:851-869); empty clusters contribute nothing
    same = torch.ones(batch, size, size)
    loss_reg = torch.zeros(1)
    for b in range(batch):
        for k in range(clusters):
            m = choice[b] == k
            if m.any():
                mean = each[b][m].mean()
                same = torch.where(torch.stack([m if bb == b else torch.zeros_like(m) for bb in range(batch)]), mean, same)
                loss_reg = loss_reg + torch.relu(mean - 0.7)
    loss_reg = loss_reg / float(batch)
    loss_tv = F.mse_loss(each, same.detach())
    attention_map = same.unsqueeze(1)
    thr = torch.where(attention_map < 0.8, attention_map - attention_map.detach(), attention_map)  # straight-through (:882-883)
    final = gaussian_blur5(thr)
    return out, final, [loss_delta, loss_reg, loss_tv], {"choice": choice, "each": each, "same": same, "pre_blur": thr}


def info_nce(image_features, clip_features, temperature=0.01):
    """run_attention.py:1315-1318: cross-entropy of the cosine-similarity matrix / 0.01 against the diagonal."""
    a = F.normalize(image_features, dim=-1)
    b = F.normalize(clip_features, dim=-1)
    sim = a @ b.T / temperature
    return F.cross_entropy(sim, torch.arange(sim.shape[0]))
