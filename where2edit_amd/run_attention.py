"""attention/run_attention.py surface, reduced to the live path: the region-attention mapper net
`FullSpaceMapperFEATClusterLinStyle_Net` (run_attention.py:703-893; the only net variant the loop can reach --
:1147,1233 pass `attention_text=`, which no other variant accepts), `GatherLayer` (utils.py:114-131), the InfoNCE term
(:1312-1318) and one iteration of `main_worker`'s loop body (:1070-1424) as `RegionAttentionTrainer.train_step`.

Same constructor arguments, parameter / buffer names (state_dict keys) and forward contract as the reference:
`forward(x, feature_map, size, attention_text=None) -> (new styles, attention map [B,1,size,size], [loss_delta, loss_reg,
loss_tv])`.  What is different by design (MI355X-first):

  * the nearest-centroid assignment is one kernel reading the cached activation in place (`w2e_cluster_assign`); the
    reference materialises position channels, a permuted copy and a [B*s*s, K, 576] broadcast temp (189 MB per sample);
  * the 18 `StyledConv(C,32,1)` + nearest resize + concat + `StyledConv(576,1,1)` + sigmoid collapse into ONE pass
    (`w2e_attention_logits`) that evaluates the 1x1 convs only at the size x size pixels the nearest resize keeps (the
    1024^2 activations are sampled at every 16th pixel instead of being convolved in full and then decimated) and never
    materialises the [B,576,size,size] concat;
  * the Python loop over B*K boolean masks (:855-868) is a per-sample reduction kernel that also applies the
    threshold and the 5x5 gaussian (`w2e_cluster_pool`).

The mask branch is forward-only: the reference keeps every `attention*` / `initial*` parameter frozen for the whole run
(:1076-1083, `t < 1.15` always holds), so nothing ever differentiates through it; if one of those parameters requires
grad while gradients are enabled this module raises instead of silently returning a constant mask.  The style branch
(`mapper_*`) is [B,1,C]-sized GEMMs on rocBLAS through torch + the fused bias/LeakyReLU op, fully differentiable."""
import ctypes

import torch
from torch import nn
from torch.nn import functional as F

from . import _lib
from ._lib import call, ptr, stream_ptr
from .stylegan2 import EqualLinear, StyledConv

_I32P = ctypes.c_void_p


class _AttSource(ctypes.Structure):  # w2e_att_source (include/w2e_attention.h)
    _fields_ = [("feat", ctypes.c_void_p), ("wscaled", ctypes.c_void_p), ("style", ctypes.c_void_p), ("demod", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("noise", ctypes.c_void_p), ("noise_w", ctypes.c_void_p), ("channels", ctypes.c_int),
                ("res", ctypes.c_int)]


PROTOS = {
    "w2e_cluster_assign": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "w2e_cluster_accumulate": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "w2e_attention_logits": (ctypes.c_int, [ctypes.POINTER(_AttSource), ctypes.c_int] + [ctypes.c_void_p] * 9 +
                             [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "w2e_attention_demod": (ctypes.c_int, [ctypes.POINTER(_AttSource), ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]),
    "w2e_cluster_pool": (ctypes.c_int, [ctypes.c_void_p] * 7 + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_void_p]),
}


def declare(lib):
    for name, (res, args) in PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args


class GLU(nn.Module):
    """utils.py:226-235"""

    def forward(self, x):
        nc = x.size(1)
        assert nc % 2 == 0, "channels dont divide 2!"
        nc = nc // 2
        return x[:, :nc] * torch.sigmoid(x[:, nc:])


class CA_NET(nn.Module):
    """utils.py:199-223.  Instantiated per edited layer by the net (`mapper_textca_{c}`, :718) but never called by its
    forward; kept so that checkpoints load strict."""

    def __init__(self, t_dim, c_dim):
        super().__init__()
        self.t_dim, self.c_dim = t_dim, c_dim
        self.fc = nn.Linear(t_dim, c_dim * 4, bias=True)
        self.relu = GLU()

    def encode(self, text_embedding):
        x = self.relu(self.fc(text_embedding))
        return x[:, :self.c_dim], x[:, self.c_dim:]

    def forward(self, text_embedding):
        mu, logvar = self.encode(text_embedding)
        std = logvar.mul(0.5).exp_()
        return torch.randn_like(std).mul(std).add_(mu), mu, logvar


def _i32ptr(t):
    if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise RuntimeError("internal: expected a contiguous int32 GPU tensor")
    return ctypes.c_void_p(t.data_ptr())


def cluster_assign(feature, centroids):
    """[B,C,s,s] activation + centroids [K, C + 2*(C//16)] -> int32 [B,s,s] nearest-centroid ids (run_attention.py:775-792;
    the x / y position channels are evaluated inside the kernel)."""
    b, c, s, s2 = feature.shape
    pos = c // 16
    if s != s2 or centroids.shape[1] != c + 2 * pos:
        raise RuntimeError(f"cluster_assign: feature {tuple(feature.shape)} vs centroids {tuple(centroids.shape)}")
    out = torch.empty((b, s, s), device=feature.device, dtype=torch.int32)
    call("w2e_cluster_assign", ptr(feature.contiguous()), ptr(centroids.contiguous()), _i32ptr(out), b, c, pos, s, centroids.shape[0],
         stream_ptr())
    return out


def cluster_pool(each, assign, size, clusters, threshold=0.8):
    """each [B,size,size] + assign int32 [B,cs,cs] -> (same [B,size,size], means [B,K], counts [B,K], thresholded
    [B,1,size,size], blurred [B,1,size,size])  (run_attention.py:843-884)."""
    b = each.shape[0]
    dev = each.device
    same = torch.empty((b, size, size), device=dev, dtype=torch.float32)
    means = torch.empty((b, clusters), device=dev, dtype=torch.float32)
    counts = torch.empty((b, clusters), device=dev, dtype=torch.float32)
    thr = torch.empty((b, 1, size, size), device=dev, dtype=torch.float32)
    final = torch.empty((b, 1, size, size), device=dev, dtype=torch.float32)
    call("w2e_cluster_pool", ptr(each.contiguous()), _i32ptr(assign), ptr(same), ptr(means), ptr(counts), ptr(thr), ptr(final), b, size,
         assign.shape[1], clusters, float(threshold), stream_ptr())
    return same, means, counts, thr, final


class FullSpaceMapperFEATClusterLinStyle_Net(nn.Module):
    """run_attention.py:703-752 (constructor) / :754-893 (forward)."""

    def __init__(self, layers, in_dim=512, latent_dim=512, attention_layer=11, cluster_layer=11, channel_multiplier=1,
                 clusters=10, cluster_dim=512):
        super().__init__()
        total_layers = layers + int((layers - 2) * 0.5)
        cm = channel_multiplier
        dim = [512] * 12 + [256 * cm] * 3 + [128 * cm] * 3 + [64 * cm] * 3 + [32 * cm] * 3 + [16 * cm] * 3
        self.layer_num = [0, 2, 3, 5, 6, 8, 9, 11, 12, 14, 15, 17, 18, 20, 21, 23, 24]
        style_layers = [0, 2, 2, 3, 5, 5, 6, 8, 8, 9, 11, 11, 12, 14, 14, 15, 17, 17, 18, 20, 20, 21, 23, 23, 24, 26, 26]
        self.mapper_layer = style_layers[attention_layer]
        for c in range(total_layers):
            if c < self.mapper_layer:
                setattr(self, f"mapper_{c}", EqualLinear(dim[c], dim[c], bias_init=1))
                setattr(self, f"mapper_textca_{c}", CA_NET(latent_dim, latent_dim))
                setattr(self, f"mapper_text_{c}", nn.Sequential(
                    EqualLinear(latent_dim, (latent_dim + 512) // 2, lr_mul=1, activation="fused_lrelu"),
                    EqualLinear((latent_dim + 512) // 2, 512, lr_mul=1, activation="fused_lrelu")))
                setattr(self, f"mapper_all_{c}", EqualLinear(dim[c] + 512, dim[c], bias_init=1))
            if c in self.layer_num:
                setattr(self, f"attention_textca_{c}", EqualLinear(latent_dim, dim[c + 1], bias_init=1))
                setattr(self, f"attention_{c}", StyledConv(dim[c + 1], 32, 1, dim[c + 1], blur_kernel=[1, 3, 3, 1]))
        self.attention_textca_first = EqualLinear(latent_dim, dim[0], bias_init=1)
        self.attention_first = StyledConv(dim[0], 32, 1, dim[0], blur_kernel=[1, 3, 3, 1])
        self.attention_textca_last = EqualLinear(latent_dim, 32 * layers, bias_init=1)
        self.attention_last = StyledConv(32 * layers, 1, 1, 32 * layers, blur_kernel=[1, 3, 3, 1])
        self.initial_bias = nn.Parameter(torch.randn(1))
        nn.init.constant_(self.initial_bias, 5)
        self.latent_dim = latent_dim
        self.register_buffer("initial_state", torch.randn(clusters, cluster_dim))
        self.cluster_layer = cluster_layer
        self.clusters = clusters

    def store_clusters(self, initial_state):
        device = self.attention_first.conv.weight.device
        assert self.initial_state.shape[0] == initial_state.shape[0], self.initial_state.shape[1] == initial_state.shape[1]
        self.initial_state = initial_state.to(device)

    # ---- the mask branch -------------------------------------------------------------------------------------------
    def _noise_is_off(self, param):
        """NoiseInjection strength == 0 (its init, and -- the mask branch being frozen -- its value for the whole run):
        answered from a cache keyed on the parameter's version, so a forward pass does not synchronise 19 times."""
        cache = self.__dict__.setdefault("_noise_off", {})
        key = (param.data_ptr(), param._version)
        hit = cache.get(id(param))
        if hit is None or hit[0] != key:
            hit = (key, bool((param.detach() == 0).all().item()))
            cache[id(param)] = hit
        return hit[1]

    def _mask_params_frozen(self):
        return not any(p.requires_grad for n, p in self.named_parameters() if n.startswith("attention") or n.startswith("initial"))

    def _wscaled_t(self, conv):
        """(scale * W[0,:,:,0,0])^T as a contiguous [C, 32] tensor, cached per conv until its weight changes."""
        cache = self.__dict__.setdefault("_wsc_t", {})
        w = conv.weight
        key = (w.data_ptr(), w._version)
        hit = cache.get(id(conv))
        if hit is None or hit[0] != key:
            hit = (key, (w.detach()[0, :, :, 0, 0] * conv.scale).t().contiguous())
            cache[id(conv)] = hit
        return hit[1]

    def _text_styles(self, mods, attention_text):
        """[aff(attention_text) for aff in mods] (the per-source style EqualLinears, :798 / :826, and attention_textca_last) from ONE
        launch of the generator's stacked-affine kernel (w2e_style_affine_fwd); None when the layers are not plain frozen
        EqualLinears of 32-multiple widths (the caller then evaluates them one by one)."""
        from . import functional as K
        if any(not isinstance(m, EqualLinear) or m.bias is None or m.activation or m.weight.shape[0] % 32 for m in mods):
            return None
        if torch.is_grad_enabled() and any(m.weight.requires_grad or m.bias.requires_grad for m in mods):
            return None
        key = tuple((m.weight.data_ptr(), m.weight._version, m.bias.data_ptr(), m.bias._version) for m in mods)
        cache = self.__dict__.get("_text_pack")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                w = torch.cat([m.weight.detach().float() * m.scale for m in mods]).contiguous()
                b = torch.cat([m.bias.detach().float() * m.lr_mul for m in mods]).contiguous()
                meta, off = [], 0
                for m in mods:
                    cw = m.weight.shape[0]
                    meta.append(torch.stack([torch.zeros(cw, dtype=torch.long), torch.full((cw,), off), torch.full((cw,), cw), torch.arange(cw)], 1))
                    off += cw
                meta = torch.cat(meta).to(device=w.device, dtype=torch.int32).contiguous()
            cache = (key, (w, b, meta, [m.weight.shape[0] for m in mods]))
            self.__dict__["_text_pack"] = cache
        return K.style_affine_all(attention_text.reshape(attention_text.shape[0], 1, -1).float(), cache[1])

    def _sources(self, n_codes):
        """(StyledConv, style EqualLinear, feature_map index) in concat order: attention_first on the const input
        (feature_map[-1], :796-802), then attention_c on feature_map[c] for the conv layers c < n_codes (:823-833)."""
        src = [(self.attention_first, self.attention_textca_first, -1)]
        src += [(getattr(self, f"attention_{c}"), getattr(self, f"attention_textca_{c}"), c) for c in self.layer_num if c < n_codes]
        return src

    @torch.no_grad()
    def attention_map(self, feature_map, size, attention_text, n_codes):
        """each_attention_map [B,size,size] (:796-842) and the nearest-centroid ids [B,cs,cs] (:763-793)."""
        batch = attention_text.shape[0]
        dev = attention_text.device
        assign = cluster_assign(feature_map[self.cluster_layer - 1], self.initial_state.to(torch.float32))
        src = self._sources(n_codes)
        if 32 * len(src) != self.attention_last.conv.in_channel:
            raise RuntimeError(f"{len(src)} attention sources for an attention_last of {self.attention_last.conv.in_channel} channels")
        descs = (_AttSource * len(src))()
        keep = []  # tensors the descriptors point into
        last = self.attention_last
        eps = src[0][0].conv.eps
        styles = None
        if attention_text.is_cuda and all(sc.conv.eps == eps for sc, _, _ in src):
            styles = self._text_styles([aff for _, aff, _ in src] + [self.attention_textca_last], attention_text)
        demods = torch.empty((len(src), batch, 32), device=dev, dtype=torch.float32) if styles is not None else None
        for j, (sc, aff, fi) in enumerate(src):
            feat = feature_map[fi]
            feat = feat if feat.is_contiguous() else feat.contiguous()
            conv = sc.conv
            if feat.shape[1] != conv.in_channel or feat.shape[2] != feat.shape[3]:
                raise RuntimeError(f"attention source {j}: feature {tuple(feat.shape)} for a {conv.in_channel}-channel conv")
            wsc = self._wscaled_t(conv)                                                 # [C,32] = (scale*W)^T, cached
            if styles is not None:  # all 19 affines came from one launch; the 18 demodulations follow in one (below)
                style, demod = styles[j], demods[j]
            else:
                style = aff(attention_text).contiguous()                               # [B,C]  (:798, :826)
                demod = torch.rsqrt(style.square() @ wsc.square() + conv.eps).contiguous()  # [B,32] (model.py:244-246)
            nw = sc.noise.weight
            noise = None if self._noise_is_off(nw) else torch.randn(batch, size * size, device=dev)  # NoiseInjection, noise=None
            bias = sc.activate.bias.contiguous()
            keep += [feat, style, wsc, demod, noise, bias]
            d = descs[j]
            d.feat, d.wscaled, d.style, d.demod, d.bias = (ptr(t).value for t in (feat, wsc, style, demod, bias))
            d.noise = ptr(noise).value if noise is not None else None
            d.noise_w = ptr(nw).value
            d.channels, d.res = conv.in_channel, feat.shape[2]
        _lib.load()
        if styles is not None:
            call("w2e_attention_demod", descs, len(src), batch, float(eps), stream_ptr())
            s_last = styles[-1]                                                         # [B, 32n]
        else:
            s_last = self.attention_textca_last(attention_text).contiguous()
        wl = (last.conv.weight[0, 0, :, 0, 0] * last.conv.scale).contiguous()           # [32n]
        d_last = torch.rsqrt((s_last * wl).square().sum(1) + last.conv.eps).contiguous()  # [B]
        nwl = last.noise.weight
        noise_last = None if self._noise_is_off(nwl) else torch.randn(batch, size * size, device=dev)
        partial = torch.empty((len(src), batch, size * size), device=dev, dtype=torch.float32)
        each = torch.empty((batch, size, size), device=dev, dtype=torch.float32)
        call("w2e_attention_logits", descs, len(src), ptr(wl), ptr(s_last), ptr(d_last), ptr(last.activate.bias.contiguous()),
             ptr(noise_last), ptr(nwl), ptr(self.initial_bias), ptr(partial), ptr(each), batch, size, stream_ptr())
        del keep
        return each, assign

    # ---- the style branch (:806-822) ----------------------------------------------------------------------------------
    def new_styles(self, x, x_text, strength_alpha=0.1):
        """`strength_alpha`: the 0.1 of :820; the demo's copy of this net takes it as an argument
        (show_demo/utils_demo.py:30: x_c + strength_alpha * (mapper_all(...) - x_c))."""
        out = []
        loss_delta = 0
        for c in range(len(x)):
            x_c = x[c][:, :, self.latent_dim:]
            if c < self.mapper_layer:
                x_text_hidden = getattr(self, f"mapper_text_{c}")(x_text).unsqueeze(1)
                x_c_hidden = getattr(self, f"mapper_{c}")(x_c)
                x_c_new = x_c + strength_alpha * (getattr(self, f"mapper_all_{c}")(torch.cat([x_c_hidden, x_text_hidden], dim=-1)) - x_c)
                loss_delta = loss_delta + torch.mean(torch.norm(x_c_new - x_c, dim=-1)) / float(self.mapper_layer)
                out.append(x_c_new.unsqueeze(3).unsqueeze(3))
            else:
                out.append(x_c.unsqueeze(3).unsqueeze(3))
        return out, loss_delta

    def forward(self, x, feature_map, size, attention_text=None, strength_alpha=0.1):
        if torch.is_grad_enabled() and not self._mask_params_frozen():
            raise RuntimeError("FullSpaceMapperFEATClusterLinStyle_Net: the mask branch (attention*/initial* parameters) is "
                               "forward-only here -- the reference keeps it frozen for the whole run (run_attention.py:1076-1083); "
                               "set requires_grad_(False) on those parameters (RegionAttentionTrainer does)")
        x_text = x[0][:, 0, :self.latent_dim]
        if attention_text is None:
            attention_text = x_text
        each, assign = self.attention_map(feature_map, size, attention_text.detach().float(), len(x))
        same, means, counts, thr, final = cluster_pool(each, assign, size, self.clusters)
        out, loss_delta = self.new_styles(x, x_text, strength_alpha)
        # :851-869: sum over non-empty clusters of relu(mean - 0.7), averaged over the batch; :871 MSE(each, same)
        loss_reg = (torch.relu(means - 0.7) * (counts > 0)).sum().reshape(1) / float(each.shape[0])
        loss_tv = F.mse_loss(each, same)
        self.last = {"each": each, "same": same, "assign": assign, "pre_blur": thr, "means": means, "counts": counts}
        return out, final, [loss_delta, loss_reg, loss_tv]


class GatherLayer(torch.autograd.Function):
    """utils.py:114-131: all_gather whose backward keeps this rank's slice of the incoming gradients (no reduction)."""

    @staticmethod
    def forward(ctx, input):
        import torch.distributed as dist
        ctx.save_for_backward(input)
        output = [torch.zeros_like(input) for _ in range(dist.get_world_size())]
        dist.all_gather(output, input.contiguous())
        return tuple(output)

    @staticmethod
    def backward(ctx, *grads):
        import torch.distributed as dist
        (input,) = ctx.saved_tensors
        grad_out = torch.zeros_like(input)
        grad_out[:] = grads[dist.get_rank()]
        return grad_out


def info_nce(image_features, clip_features, temperature=0.01):
    """run_attention.py:1312-1318: gather both feature sets over the ranks, then cross-entropy of the cosine-similarity
    matrix / 0.01 against the diagonal."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        image_features = torch.cat(GatherLayer.apply(image_features), dim=0)
        clip_features = torch.cat(GatherLayer.apply(clip_features), dim=0)
    a = F.normalize(image_features, dim=-1)
    b = F.normalize(clip_features, dim=-1)
    sim = a @ b.T / temperature
    return F.cross_entropy(sim, torch.arange(sim.shape[0], device=sim.device))


def get_lr(t, initial_lr, rampdown=0.25, rampup=0.05):
    """run_attention.py:37-42"""
    import math
    lr_ramp = min(1, (1 - t) / rampdown)
    lr_ramp = 0.5 - 0.5 * math.cos(lr_ramp * math.pi)
    lr_ramp = lr_ramp * min(1, t / rampup)
    return initial_lr * lr_ramp


class RegionAttentionTrainer:
    """One iteration of `main_worker`'s loop (run_attention.py:1070-1424) in the shipped configuration
    (attention/train_scripts.sh:3: --work_in_stylespace --use_cluster, attention_layer = cluster_layer = 13, K = 20,
    batch 1 per GPU on 8 GPUs, Adam).  Per step and rank:

        G(w1) with features (no grad)          -> img_orig                      (:1090-1104)
        CLIP image features of img_orig        -> clip_features_origin          (:1163-1172)
        G(w2) with features (no grad)          -> the "consistency" sample      (:1189-1205)
        sample 0 of rank 0 replaces the batch  (X3-X5, :1208-1230)
        Mapper(clip features (+) S codes, features) -> new S codes, mask        (:1231-1240)
        G(new codes, blend at attention_layer under the mask)  [autograd]       (:1245)
        CLIP image features -> InfoNCE over the gathered global batch           (:1259-1260, :1312-1318)
        loss_total -> backward -> mean all-reduce of the mapper gradients -> Adam (:1415-1424)

    `consistency`: "recompute" (default here) broadcasts rank 0's W+ latent of sample 0 (36 KB) and re-runs the generator
    on it locally; "broadcast" is the reference's pattern (27 feature maps + 26 codes + the image = 542 MB per step from
    rank 0).  Both give every rank the same tensors.  `identity_loss`: None (lambda_id term off) or a module
    `(img_gen, img_orig) -> (loss, _)` such as IDLoss -- the reference's VGG perceptual loss needs torchvision weights
    and is out of scope.  Text prompts: the reference samples phrases and tokenises them with OpenAI's BPE
    (not in this image); callers pass CLIP text features / token ids."""

    def __init__(self, g_ema, clip_loss, mapper, *, attention_layer=13, lr=0.01, steps=10000, lambda_ess=0.03, lambda_sec=0.01,
                 lambda_id=0.1, lambda_delta=0.03, identity_loss=None, consistency="recompute", device="cuda:0", amp=False):
        from . import dist as w2e_dist
        self.device = device
        self.g_ema = g_ema.to(device).eval().requires_grad_(False)
        self.clip_loss = clip_loss.to(device)
        self.mapper = mapper.to(device)
        for n, p in self.mapper.named_parameters():  # run_attention.py:1076-1083 (t < 1.15 always holds)
            if n.startswith("attention") or n.startswith("initial"):
                p.requires_grad_(False)
        self.attention_layer = attention_layer
        self.lr, self.steps = lr, steps
        self.lambdas = (lambda_ess, lambda_sec, lambda_id, lambda_delta)
        self.identity_loss = identity_loss.to(device) if identity_loss is not None else None
        if consistency not in ("recompute", "broadcast"):
            raise ValueError("consistency must be 'recompute' or 'broadcast'")
        self.consistency = consistency
        self.params = [p for p in self.mapper.parameters() if p.requires_grad]
        self.optimizer = torch.optim.Adam(self.params, lr=lr)
        # `--amp` (run_attention.py:1068-1069, 1231, 1418-1421): the reference wraps the mapper + generator forward in autocast and drives
        # the optimizer through a GradScaler.  The kernels of this package compute in fp32 only, so there is nothing to autocast (a
        # narrower forward would also leave the north_star tolerance); what `amp=True` keeps is the GradScaler protocol -- the loss is
        # scaled before backward, the (all-reduced) gradients are unscaled and checked, a step with a non-finite gradient is SKIPPED
        # and the scale backs off -- so a run configured with --amp keeps its skip-on-overflow behaviour.  The scale is a power of two:
        # with finite gradients the parameters after a step are bit-identical to amp=False.
        self.amp = bool(amp)
        self.scaler = torch.amp.GradScaler("cuda", enabled=self.amp)
        import torch.distributed as dist
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.bucket = w2e_dist.GradBucket(self.params) if self.world > 1 else None
        self.global_step = 0

    # ---- pieces ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _generate(self, w):
        img, _, codes, feats = self.g_ema([w], input_is_latent=True, randomize_noise=False, return_features=True)
        feats = list(feats)
        feats.append(self.g_ema.input.input.repeat(w.shape[0], 1, 1, 1))  # :1110 / :1205
        return img, [s.detach() for s in codes], feats

    def _encode_image(self, img):
        return self.clip_loss.model.encode_image(self.clip_loss.preprocess(img))

    def _first_sample(self, w2, img, codes, feats, batch):
        """X3-X5: every rank continues with sample 0 of rank 0, repeated `batch` times."""
        import torch.distributed as dist
        if self.consistency == "recompute":
            w0 = w2[:1].clone()
            if self.world > 1:
                dist.broadcast(w0, 0)
            if self.world > 1 or batch > 1:
                img, codes, feats = self._generate(w0)
            return (img[:1].repeat(batch, 1, 1, 1), [s[:1].repeat(batch, 1, 1, 1, 1) for s in codes],
                    [f[:1].repeat(batch, 1, 1, 1) for f in feats])
        first = []
        for f in feats:
            t = f[:1].clone()
            if self.world > 1:
                dist.broadcast(t, 0)
            first.append(t.repeat(batch, 1, 1, 1))
        first_codes = []
        for s in codes:
            t = s[:1].clone()
            if self.world > 1:
                dist.broadcast(t, 0)
            first_codes.append(t.repeat(batch, 1, 1, 1, 1))
        t = img[:1].clone()
        if self.world > 1:
            dist.broadcast(t, 0)
        return t.repeat(batch, 1, 1, 1), first_codes, first

    def losses(self, w1, w2, attention_text_features):
        """Everything up to `loss_total` for this rank's latents w1, w2 [B,18,512] (the two fresh batches of :1090 and
        :1189) and the attention prompt's CLIP text features [B,512] (:1139).  Returns (loss_total, dict, img_gen)."""
        import torch.distributed as dist
        batch = w1.shape[0]
        t = self.global_step / self.steps
        img_orig, _, _ = self._generate(w1)
        with torch.no_grad():
            clip_features_origin = self._encode_image(img_orig)  # :1163-1172 (clip_features_origin = image_features_origin)
            first_text = attention_text_features[:1].float().clone()
            if self.world > 1:
                dist.broadcast(first_text, 0)
            first_text = first_text.repeat(batch, 1)
        img2, codes2, feats2 = self._generate(w2)
        first_img, first_codes, first_feats = self._first_sample(w2, img2, codes2, feats2, batch)
        blend_size = first_feats[self.attention_layer - 1].shape[-1]
        x = [torch.cat([clip_features_origin.unsqueeze(1), s[:, :, :, 0, 0]], dim=-1) for s in first_codes]  # :1240
        new_codes, attention_map, delta_loss = self.mapper(x, first_feats, blend_size, attention_text=first_text)
        img_gen, _ = self.g_ema([new_codes], input_is_latent=True, randomize_noise=False, input_is_stylespace=True,
                                attention_layer=self.attention_layer, attention_map=attention_map, feature_map=first_feats)
        image_features = self._encode_image(img_gen)
        loss_consist = info_nce(image_features, clip_features_origin)
        loss_delta, loss_sec, loss_ess = delta_loss[0], delta_loss[1], delta_loss[2]
        l_ess, l_sec, l_id, l_delta = self.lambdas
        ramp1 = max(0, min(1, (t - 0.15) / 0.1))
        ramp2 = max(0, min(1, (t - 0.05) / 0.1))
        total = loss_consist + ramp1 * (l_ess * loss_ess + l_sec * loss_sec.squeeze()) + l_delta * loss_delta
        d = {"loss_consist": loss_consist.detach(), "loss_essence": loss_ess.detach(), "loss_secphase": loss_sec.detach(),
             "loss_delta": loss_delta.detach()}
        if self.identity_loss is not None:
            loss_identity = self.identity_loss(img_gen, first_img)[0]
            total = total + ramp2 * (l_id * loss_identity)
            d["loss_identity"] = loss_identity.detach()
        d["loss"] = total.detach()
        return total, d, img_gen

    def train_step(self, w1, w2, attention_text_features):
        self.mapper.train()
        t = self.global_step / self.steps
        self.optimizer.param_groups[0]["lr"] = get_lr(t, self.lr)  # :1072-1074
        if self.bucket is not None:
            self.bucket.zero()
        else:
            self.optimizer.zero_grad()
        total, d, _ = self.losses(w1, w2, attention_text_features)
        self.scaler.scale(total).backward()  # (:1418; the identity when amp is off)
        if self.bucket is not None:
            self.bucket.all_reduce_mean()  # DDP's averaged gradient (X1), one flat message (of the scaled gradients: every rank holds the same scale)
        self.scaler.step(self.optimizer)   # (:1419-1420: unscale, skip the step on inf / nan)
        self.scaler.update()               # (:1421)
        self.global_step += 1
        return d

    @torch.no_grad()
    def sample_latents(self, batch, mean_latent, truncation=0.7, generator=None):
        """:1090-1093: z ~ N(0,I) -> style MLP -> truncation toward mean_latent -> W+."""
        z = torch.randn(batch, 512, generator=generator).to(self.device)
        w = mean_latent + truncation * (self.g_ema.style(z) - mean_latent)
        return w.unsqueeze(1).repeat(1, self.g_ema.n_latent, 1).contiguous()
