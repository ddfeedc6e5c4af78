"""N2: the region-attention mapper net (attention/run_attention.py:703-893) on the HIP kernels against (a) the fixture
captured from the reference's own class (tests/golden/attention_net.npz, make_golden_attention.py) and (b) the CPU oracle at
FFHQ-1024 shapes."""
import numpy as np
import pytest
import torch

import make_golden_attention as M
import seeded
from helpers import assert_close, golden
from oracle import attention_net as OA

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net():
    from where2edit_amd.run_attention import FullSpaceMapperFEATClusterLinStyle_Net
    net = FullSpaceMapperFEATClusterLinStyle_Net(M.LAYERS, 1024, 512, attention_layer=M.ATT_LAYER, channel_multiplier=2,
                                                 cluster_layer=M.CLUSTER_LAYER, clusters=M.CLUSTERS, cluster_dim=576)
    sd = M.net_state_dict(net)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).train()
    for n, p in net.named_parameters():  # the reference's schedule: mask branch frozen (run_attention.py:1076-1083)
        if n.startswith("attention") or n.startswith("initial"):
            p.requires_grad_(False)
    return net, sd


def test_cluster_assign_matches_reference_fixture():
    from where2edit_amd.run_attention import cluster_assign
    g = golden("attention_net")
    feats = M.feature_maps()
    a = cluster_assign(feats[M.CLUSTER_LAYER - 1].to(DEV), M.centroids().to(DEV))
    assert a.dtype == torch.int32 and torch.equal(a.cpu(), torch.from_numpy(g["assign"]))


def test_region_attention_net_matches_reference_fixture():
    g = golden("attention_net")
    net, _ = _net()
    x, att_text, _ = M.inputs()
    feats = [f.to(DEV) for f in M.feature_maps()]
    out, final, losses = net([t.to(DEV) for t in x], feats, M.SIZE, attention_text=att_text.to(DEV))
    assert len(out) == M.LAYERS + (M.LAYERS - 2) // 2
    for c, o in enumerate(out):
        assert_close(o, g[f"out{c}"], 1e-5, f"new style {c}")
    assert_close(net.last["pre_blur"], g["pre_blur"], 1e-4, "thresholded map (what the reference hands to gaussian_blur)")
    assert_close(final, g["final_map"], 1e-4, "blurred map")
    for name, val in zip(("loss_delta", "loss_reg", "loss_tv"), losses):
        assert abs(float(val.detach()) - float(g[name])) <= 1e-4 * max(abs(float(g[name])), 1e-3), name
    r = [seeded.tensor(f"att.r{c}", tuple(o.shape)).to(DEV) for c, o in enumerate(out)]
    scalar = sum((o * rr).sum() for o, rr in zip(out, r)) + 3.0 * losses[0]
    names = [str(n) for n in g["grad_names"]]
    params = dict(net.named_parameters())
    grads = torch.autograd.grad(scalar, [params[n] for n in names])
    for n, gg in zip(names, grads):
        assert_close(gg, g["grad." + n], 1e-4, "grad " + n)


def test_mask_branch_refuses_to_be_trained():
    net, _ = _net()
    net.attention_first.activate.bias.requires_grad_(True)
    x, att_text, _ = M.inputs()
    with pytest.raises(RuntimeError, match="forward-only"):
        net([t.to(DEV) for t in x], [f.to(DEV) for f in M.feature_maps()], M.SIZE, attention_text=att_text.to(DEV))


def test_attention_map_at_ffhq1024_shapes_vs_oracle():
    """The shipped setting (attention/train_scripts.sh:3): 1024^2 generator (18 W+ / 26 S codes), attention_layer = cluster_layer
    = 13 (64x64), K = 20 clusters, 576-D centroids -- every source resolution 4..1024 through the nearest gather, the 160-KB
    LDS opt-in of the assignment kernel, batch 2.  Features are seeded tensors of the generator's shapes."""
    from where2edit_amd.run_attention import FullSpaceMapperFEATClusterLinStyle_Net
    layers, att, k, size, b = 18, 13, 20, 64, 2
    res = [4, 4] + [r for r in (8, 16, 32, 64, 128, 256, 512, 1024) for _ in range(3)]
    ch = [512, 3] + [c for c in (512, 512, 512, 512, 256, 128, 64, 32) for c in (c, c, 3)]
    feats = [seeded.tensor(f"att1024.f{i}", (b, c, r, r)) for i, (r, c) in enumerate(zip(res, ch))]
    protos = seeded.tensor("att1024.protos", (k, 512), 1.0)
    lab = torch.from_numpy(np.random.RandomState(5).randint(0, k - 2, size=(b, 8, 8))).repeat_interleave(8, 1).repeat_interleave(8, 2)
    feats[att - 1] = (protos[lab].permute(0, 3, 1, 2) + 0.25 * seeded.tensor("att1024.noise", (b, 512, 64, 64))).contiguous()
    feats.append(seeded.tensor("att1024.const", (1, 512, 4, 4)).repeat(b, 1, 1, 1))
    net = FullSpaceMapperFEATClusterLinStyle_Net(layers, 1024, 512, attention_layer=att, channel_multiplier=2, cluster_layer=att,
                                                 clusters=k, cluster_dim=576)
    sd = M.net_state_dict(net)
    sd["initial_state"] = torch.cat([protos, 0.05 * seeded.tensor("att1024.cpos", (k, 64))], 1)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).requires_grad_(False)
    text = seeded.tensor("att1024.text", (b, 512), 0.3)
    att_text = seeded.tensor("att1024.att_text", (1, 512), 0.3).repeat(b, 1)
    dims = OA.dims(2)
    x = [torch.cat([text.unsqueeze(1), seeded.tensor(f"att1024.s{c}", (b, 1, dims[c]), 0.5, 1.0)], -1) for c in range(26)]
    with torch.no_grad():
        out_o, final_o, losses_o, extra = OA.forward(sd, x, feats, size, attention_text=att_text, attention_layer=att, cluster_layer=att,
                                                     clusters=k)
        out, final, losses = net([t.to(DEV) for t in x], [f.to(DEV) for f in feats], size, attention_text=att_text.to(DEV))
    assert torch.equal(net.last["assign"].cpu().long(), extra["choice"])
    assert_close(net.last["each"], extra["each"], 1e-4, "each_attention_map")
    assert_close(net.last["same"], extra["same"], 1e-4, "cluster-pooled map")
    assert_close(final, final_o, 1e-4, "final map")
    for a_, b_ in zip(out, out_o):
        assert_close(a_, b_, 1e-5, "new style")
    for a_, b_, name in zip(losses, losses_o, ("loss_delta", "loss_reg", "loss_tv")):
        assert abs(float(a_) - float(b_)) <= 1e-4 * max(abs(float(b_)), 1e-3), name


def _trainer(size=256, consistency="recompute", identity=False, amp=False):
    import types
    from make_golden import CLIP_TINY as c
    from where2edit_amd.attention_model import Generator
    from where2edit_amd.clip_loss import CLIPLoss
    from where2edit_amd.clip_vit import CLIP
    from where2edit_amd.run_attention import FullSpaceMapperFEATClusterLinStyle_Net, RegionAttentionTrainer
    gsd = seeded.generator_state_dict(size)
    g = Generator(size, 512, 8)
    g.load_state_dict(gsd, strict=True)
    clip = CLIP(embed_dim=c["embed_dim"], vision_layers=c["vision_layers"], vision_width=c["vision_width"],
                context_length=c["context_length"], vocab_size=c["vocab_size"], transformer_width=c["text_width"],
                transformer_heads=1, transformer_layers=c["text_layers"])
    csd = seeded.clip_state_dict(**c)
    clip.load_state_dict(csd, strict=True)
    net = FullSpaceMapperFEATClusterLinStyle_Net(M.LAYERS, c["embed_dim"] + 512, c["embed_dim"], attention_layer=M.ATT_LAYER,
                                                 channel_multiplier=2, cluster_layer=M.CLUSTER_LAYER, clusters=M.CLUSTERS, cluster_dim=576)
    msd = M.net_state_dict(net)
    net.load_state_dict(msd, strict=True)
    opts = types.SimpleNamespace(stylegan_size=size)
    tr = RegionAttentionTrainer(g, CLIPLoss(opts, model=clip), net, attention_layer=M.ATT_LAYER, lr=0.01, steps=100,
                                consistency=consistency, device=DEV, amp=amp)
    return tr, gsd, csd, msd, c["embed_dim"]


def test_region_attention_trainer_step_matches_oracle():
    """One iteration of main_worker's loop body (run_attention.py:1070-1424; S-space + clusters, single process) against
    the oracle composition: G with features -> CLIP image features -> net -> masked G -> CLIP -> InfoNCE + lambda_delta *
    loss_delta; loss terms and the gradients of every trainable (mapper_*) parameter; Adam moves only those."""
    from oracle import clip_model as OC
    from oracle import ops as OO
    from oracle import stylegan2 as OG
    size, b = 256, 2
    tr, gsd, csd, msd, edim = _trainer(size)
    tr.global_step = 30  # t = 0.3: both ramps are past their start (:1415)
    w1 = seeded.wplus_latents(b, OG.n_latent(size), salt=51)
    w2 = seeded.wplus_latents(b, OG.n_latent(size), salt=52)
    att_text = seeded.tensor("trainer.att_text", (b, edim), 0.3)
    # oracle
    names = [n for n, p in tr.mapper.named_parameters() if p.requires_grad]
    assert names and all(n.startswith("mapper_") for n in names)
    osd = {k: v.clone() for k, v in msd.items()}
    for n in names:
        osd[n].requires_grad_(True)
    with torch.no_grad():
        img1, _, _, _ = OG.generator_forward(gsd, [w1], size=size, input_is_latent=True, randomize_noise=False, return_features=True)
        cfo = OC.encode_image(csd, OO.clip_preprocess(img1, size))
        img2, _, codes2, feats2 = OG.generator_forward(gsd, [w2], size=size, input_is_latent=True, randomize_noise=False, return_features=True)
        feats2 = list(feats2) + [gsd["input.input"].repeat(b, 1, 1, 1)]
        first_feats = [f[:1].repeat(b, 1, 1, 1) for f in feats2]
        first_codes = [s[:1].repeat(b, 1, 1, 1, 1) for s in codes2]
    x = [torch.cat([cfo.unsqueeze(1), s[:, :, :, 0, 0]], -1) for s in first_codes]
    first_text = att_text[:1].repeat(b, 1)
    new_codes, amap, dl, _ = OA.forward(osd, x, first_feats, M.SIZE, attention_text=first_text, attention_layer=M.ATT_LAYER,
                                        cluster_layer=M.CLUSTER_LAYER, clusters=M.CLUSTERS, latent_dim=edim)
    img_gen, _ = OG.generator_forward(gsd, [new_codes], size=size, input_is_stylespace=True, randomize_noise=False,
                                      attention_layer=M.ATT_LAYER, attention_map=amap, feature_map=first_feats)
    feat_gen = OC.encode_image(csd, OO.clip_preprocess(img_gen, size))
    l_consist = OA.info_nce(feat_gen, cfo)
    total_o = l_consist + 1.0 * (0.03 * dl[2] + 0.01 * dl[1].squeeze()) + 0.03 * dl[0]
    grads_o = torch.autograd.grad(total_o, [osd[n] for n in names], allow_unused=True)
    unused = [n for n, g in zip(names, grads_o) if g is None]
    assert unused and all("mapper_textca_" in n for n in unused)  # CA_NETs are constructed but never called (:718, :808-810)
    # HIP
    before = {n: p.detach().clone() for n, p in tr.mapper.named_parameters()}
    d = tr.train_step(w1.to(DEV), w2.to(DEV), att_text.to(DEV))
    for key, ref in (("loss_consist", l_consist), ("loss_delta", dl[0]), ("loss_secphase", dl[1]), ("loss_essence", dl[2]), ("loss", total_o)):
        assert abs(float(d[key]) - float(ref.detach())) <= 2e-4 * max(abs(float(ref.detach())), 1e-3), (key, float(d[key]), float(ref.detach()))
    params = dict(tr.mapper.named_parameters())
    from helpers import assert_grad_close
    assert all(params[n].grad is None for n in unused)
    used = [(n, g) for n, g in zip(names, grads_o) if g is not None]
    assert_grad_close(torch.cat([params[n].grad.reshape(-1).cpu() for n, _ in used]), torch.cat([g.reshape(-1) for _, g in used]),
                      "trainable mapper parameters")
    moved = [n for n, p in params.items() if not torch.equal(p.detach(), before[n])]
    assert moved and all(n.startswith("mapper_") for n in moved)


def test_region_attention_trainer_amp_keeps_the_gradscaler_protocol():
    """`amp=True` (the reference's --amp: run_attention.py:1068-1069, 1418-1421): the loss is scaled before backward, scaler.step unscales
    and steps -- the scale is a power of two, so with finite gradients three steps leave the parameters bit-identical to amp=False -- and
    a step whose gradient is not finite is SKIPPED (parameters untouched) while the scale backs off."""
    import where2edit_amd
    lat = lambda salt: seeded.wplus_latents(1, 14, salt=salt).to(DEV)  # noqa: E731  (Generator(256): 14 latents)
    runs = []
    where2edit_amd.set_deterministic(True)  # (no fp32 atomics: two runs of the same step are bit-identical, so amp on / off can be compared exactly)
    try:
        for amp in (False, True):
            tr, _, _, _, edim = _trainer(amp=amp)
            text = seeded.tensor("amp.att", (1, edim), 0.3).to(DEV)
            for i in range(3):
                tr.train_step(lat(200 + i), lat(300 + i), text)
            runs.append(tr)
    finally:
        where2edit_amd.set_deterministic(False)
    a, b = runs
    assert b.scaler.is_enabled() and not a.scaler.is_enabled() and b.scaler.get_scale() == 65536.0
    for (n, pa), (_, pb) in zip(a.mapper.named_parameters(), b.mapper.named_parameters()):
        assert torch.equal(pa, pb), n
    before = {n: p.detach().clone() for n, p in b.mapper.named_parameters()}
    poisoned = next(p for p in b.params)
    h = poisoned.register_hook(lambda g: g * float("inf"))  # an overflowing backward
    b.train_step(lat(210), lat(310), text)
    h.remove()
    assert all(torch.equal(before[n], p.detach()) for n, p in b.mapper.named_parameters()), "a non-finite step was applied"
    assert b.scaler.get_scale() == 32768.0


def test_gpu_lloyd_matches_cpu_lloyd():
    """attention/clustering_feature.py:212-235 on the GPU kernels (assignment + fixed-order per-cluster sums on the
    up-sampled activation in place) against a literal CPU Lloyd on the [N,576] point matrix, same initial centres."""
    from where2edit_amd import clustering_feature as CF
    b, c, s, k = 2, 512, 16, 6
    protos = seeded.tensor("kmeans.protos", (k, c), 1.0)
    lab = torch.from_numpy(np.random.RandomState(3).randint(0, k, size=(b, s, s)))
    feat = (protos[lab].permute(0, 3, 1, 2) + 0.3 * seeded.tensor("kmeans.noise", (b, c, s, s))).contiguous()
    pts = CF.clustering_points(feat.to(DEV))                     # [B,512,32,32]
    X = CF.points_matrix(pts).cpu()                              # literal [N,576]
    assert_close(X, CF.points_matrix(torch.nn.functional.interpolate(feat, size=2 * s, mode="bilinear", align_corners=True)), 1e-6)
    init = X[torch.tensor([5, 300, 700, 1100, 1500, 1900])].clone()
    cen = init.clone()
    for _ in range(300):  # the reference's loop
        ch = torch.argmin(OA.pairwise_distance(X, cen), 1)
        new = torch.stack([X[ch == i].mean(0) if (ch == i).any() else cen[i] for i in range(k)])
        shift = torch.sum(torch.sqrt(torch.sum((new - cen) ** 2, 1)))
        cen = new
        if shift ** 2 < 1e-4:
            break
    assign, centres = CF.lloyd(pts, k, initial_state=init)
    assert torch.equal(assign.cpu().long().reshape(-1), ch)
    assert_close(centres, cen, 1e-5, "k-means centres")
    sums, counts = CF.cluster_sums(pts, assign, k)
    assert torch.equal(counts.cpu().long(), torch.bincount(ch, minlength=k))
