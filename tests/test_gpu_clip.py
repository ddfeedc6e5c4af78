"""CLIP image-encoder kernels (GEMM / LayerNorm / attention) and the CLIP loss against the CPU oracle and the
transformers cross-check fixture.  OpenAI `clip` is absent from the reference tree: reference parity is
UNPINNED for this component (see oracle/clip_model.py); tolerance is stated against the fp32 restatement."""
import pytest
import torch

import seeded
from helpers import assert_close, golden, rel_err
from make_golden import CLIP_TINY
from oracle import clip_model as OC

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(cfg, text=True):
    from where2edit_amd.clip_vit import CLIP
    m = CLIP(embed_dim=cfg["embed_dim"], image_resolution=cfg["image_resolution"], vision_layers=cfg["vision_layers"],
             vision_width=cfg["vision_width"], vision_patch_size=cfg["vision_patch"], context_length=cfg["context_length"],
             vocab_size=cfg["vocab_size"], transformer_width=cfg["text_width"], transformer_heads=cfg["text_width"] // 64,
             transformer_layers=cfg["text_layers"])
    sd = seeded.clip_state_dict(**cfg)
    m.load_state_dict(sd, strict=True)  # OpenAI key layout loads unchanged
    for p in m.parameters():
        p.requires_grad_(False)
    return m.to(DEV).eval(), sd


@pytest.mark.parametrize("m,n,k,trans_b", [(200, 2304, 768, True), (200, 768, 3072, True), (196, 768, 3072, True),
                                            (200, 768, 2304, False), (37, 64, 32, True), (4, 512, 768, True), (130, 192, 68, False)])
def test_gemm_vs_torch(m, n, k, trans_b):
    from where2edit_amd import vit_hip
    a = torch.randn(m, k, device=DEV)
    b = torch.randn(n, k, device=DEV) if trans_b else torch.randn(k, n, device=DEV)
    bias = torch.randn(n, device=DEV)
    res = torch.randn(m, n, device=DEV)
    ref = (a.double() @ (b.double().t() if trans_b else b.double())) + bias.double() + res.double()
    c = vit_hip._gemm(a, b, trans_b, bias=bias, residual=res)
    assert_close(c, ref, 2e-6 * (k ** 0.5))
    if trans_b:
        g = a.double() * torch.sigmoid(1.702 * a.double())
        c2 = vit_hip._gemm(a, b, True, a_gelu=True)
        assert_close(c2, g @ b.double().t(), 1e-5)
    aux = torch.randn(m, n, device=DEV)
    s = torch.sigmoid(1.702 * aux.double())
    c3 = vit_hip._gemm(a, b, trans_b, gelu_grad_aux=aux)
    assert_close(c3, (a.double() @ (b.double().t() if trans_b else b.double())) * (s * (1 + 1.702 * aux.double() * (1 - s))), 1e-5)


def test_layernorm_and_attention_vs_torch():
    from where2edit_amd import vit_hip
    x = torch.randn(3, 50, 768, device=DEV, requires_grad=True)
    ln = torch.nn.LayerNorm(768).to(DEV)
    with torch.no_grad():
        ln.weight.normal_(1, 0.2), ln.bias.normal_(0, 0.2)
    ref = torch.nn.functional.layer_norm(x, (768,), ln.weight, ln.bias, 1e-5)
    g = torch.randn_like(ref)
    (gref,) = torch.autograd.grad(ref, x, g)
    ln.requires_grad_(False)
    y = vit_hip.layer_norm(x, ln)
    (gx,) = torch.autograd.grad(y, x, g)
    assert_close(y, ref, 1e-5), assert_close(gx, gref, 1e-5)
    for L, H in ((50, 12), (7, 2), (64, 1)):
        qkv = torch.randn(2, L, 3 * H * 64, device=DEV, requires_grad=True)
        q, k, v = qkv.view(2, L, 3, H, 64).permute(2, 0, 3, 1, 4)
        ref = ((q @ k.transpose(-1, -2) / 8).softmax(-1) @ v).transpose(1, 2).reshape(2, L, H * 64)
        g = torch.randn_like(ref)
        (gref,) = torch.autograd.grad(ref, qkv, g)
        out = vit_hip._Attention.apply(qkv, H)
        (gq,) = torch.autograd.grad(out, qkv, g)
        assert_close(out, ref, 1e-5, f"attn L={L}"), assert_close(gq, gref, 2e-5, f"attn grad L={L}")


def test_clip_tiny_logits_and_image_grad_vs_oracle_and_transformers():
    g = golden("clip_hf")
    m, sd = _model(CLIP_TINY)
    img = seeded.tensor("clip.tiny.img", (3, 3, 224, 224), 0.5)
    tokens = torch.from_numpy(g["tiny.tokens"])
    ig = img.to(DEV).requires_grad_(True)
    logits, logits_t = m(ig, tokens.to(DEV))
    assert_close(logits, g["tiny.logits_per_image"], 1e-4, "vs transformers")
    assert logits_t.shape == (2, 3)
    io = img.clone().requires_grad_(True)
    lo = OC.clip_logits(sd, io, tokens)
    assert_close(logits, lo, 1e-4, "vs oracle")
    (gi,) = torch.autograd.grad(logits.sum(), ig)
    (go,) = torch.autograd.grad(lo.sum(), io)
    assert_close(gi, go, 1e-3, "image gradient")


def test_vit_b32_visual_features():
    g = golden("clip_hf")
    cfg = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch=32, context_length=8,
               vocab_size=64, text_width=64, text_layers=1)
    m, _ = _model(cfg)
    f = m.encode_image(seeded.tensor("clip.b32.img", (2, 3, 224, 224), 0.5).to(DEV))
    assert_close(f, g["b32.image_features"], 1e-3, "ViT-B/32 image features (north_star tolerance)")


def test_clip_loss_1024_vs_oracle():
    """CLIPLoss.forward at the benchmark image size: fused preprocessing + ViT-B/32 + cached text features."""
    import types
    from where2edit_amd.clip_loss import CLIPLoss
    from where2edit_amd.coach import synthetic_tokens
    cfg = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch=32, context_length=77,
               vocab_size=49408, text_width=512, text_layers=12)
    m, sd = _model(cfg)
    loss = CLIPLoss(types.SimpleNamespace(stylegan_size=1024), model=m).to(DEV)
    assert isinstance(loss.upsample, torch.nn.Upsample) and loss.avg_pool.kernel_size == 32
    img = seeded.tensor("cliploss.img", (1, 3, 1024, 1024), 0.5)
    tokens = synthetic_tokens(2)
    ig = img.to(DEV).requires_grad_(True)
    out = loss(ig, tokens.to(DEV))
    io = img.clone().requires_grad_(True)
    ref = OC.clip_loss(sd, io, tokens, 1024)
    assert out.shape == (1, 2)
    assert_close(out, ref, 1e-4)
    (gi,) = torch.autograd.grad(out.sum(), ig)
    (go,) = torch.autograd.grad(ref.sum(), io)
    assert_close(gi, go, 2e-3, "d loss / d image")
    assert loss(ig, tokens.to(DEV)) is not None and m._text_cache is not None


# ---------------------------------------------------------------------------------------------- the M = 50*batch kernels
def test_reduce_ln_and_attention_v2_vs_torch():
    from where2edit_amd import vit_hip as V
    from where2edit_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(3)
    for dim in (768, 512, 1024):
        m = 150
        part = torch.randn(3, m, dim, generator=g).to(DEV)
        bias, res = torch.randn(dim, generator=g).to(DEV), torch.randn(m, dim, generator=g).to(DEV)
        gamma, beta = (torch.randn(dim, generator=g) * 0.2 + 1).to(DEV), (torch.randn(dim, generator=g) * 0.2).to(DEV)
        x, y, mean, rstd = V._reduce_ln(part, bias, res, gamma, beta, 1e-5)
        xr = (part.double().sum(0) + bias.double() + res.double()).requires_grad_(True)
        yr = torch.nn.functional.layer_norm(xr, (dim,), gamma.double(), beta.double(), 1e-5)
        assert_close(x, xr, 1e-6, "reduced x"), assert_close(y, yr, 1e-5, "LayerNorm of the reduced x")
        gp = torch.randn(2, m, dim, generator=g).to(DEV)
        add = torch.randn(m, dim, generator=g).to(DEV)
        (gref,) = torch.autograd.grad(yr, xr, gp.double().sum(0))
        assert_close(V._ln_bwd_part(gp, x, gamma, mean, rstd, add), gref + add.double(), 1e-5, "LN backward of summed slabs + add")
        x2, y2, _, _ = V._reduce_ln(part, None, None, None, None, 0.0, want_y=False)
        assert y2 is None and torch.allclose(x2.double(), part.double().sum(0), atol=1e-5)
    for L, H, B, S in ((50, 12, 4, 3), (7, 2, 2, 1), (64, 1, 1, 2), (33, 3, 2, 1)):
        d3 = 3 * H * 64
        slabs = torch.randn(S, B * L, d3, generator=g).to(DEV)
        bias = torch.randn(d3, generator=g).to(DEV)
        qkv = (slabs.double().sum(0) + bias.double()).view(B, L, d3).requires_grad_(True)
        q, k, v = qkv.view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
        ref = ((q @ k.transpose(-1, -2) / 8).softmax(-1) @ v).transpose(1, 2).reshape(B * L, H * 64)
        out = torch.empty(B * L, H * 64, device=DEV)
        call("w2e_attn2_fwd", ptr(slabs), S, B * L * d3, ptr(bias), ptr(out), B, L, H, 0, stream_ptr())
        assert_close(out, ref, 1e-5, f"attn2 L={L}")
        gs = torch.randn(2, B * L, H * 64, generator=g).to(DEV)
        (gref,) = torch.autograd.grad(ref, qkv, gs.double().sum(0))
        gq = torch.empty(B * L, d3, device=DEV)
        call("w2e_attn2_bwd", ptr(slabs), S, B * L * d3, ptr(bias), ptr(gs), 2, B * L * H * 64, ptr(gq), B, L, H, 0, stream_ptr())
        assert_close(gq, gref.reshape(B * L, d3), 2e-5, f"attn2 grad L={L}")


def test_vit_b32_tower_v2_matches_first_generation_and_oracle_gradient(monkeypatch):
    """The full ViT-B/32 visual tower on the M = 50*batch kernels: features and image gradient against the first-generation
    kernels (W2E_VIT_V1) and against the CPU oracle, batch 4 (M = 200) and batch 5 (two M tiles)."""
    cfg = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch=32, context_length=8,
               vocab_size=64, text_width=64, text_layers=1)
    m, sd = _model(cfg)
    for b in (4, 5):
        img = seeded.tensor(f"clip.v2.img{b}", (b, 3, 224, 224), 0.5)
        r = seeded.tensor(f"clip.v2.r{b}", (b, 512))
        ig = img.to(DEV).requires_grad_(True)
        f2 = m.encode_image(ig)
        (g2,) = torch.autograd.grad((f2 * r.to(DEV)).sum(), ig)
        monkeypatch.setenv("W2E_VIT_V1", "1")
        i1 = img.to(DEV).requires_grad_(True)
        f1 = m.encode_image(i1)
        (g1,) = torch.autograd.grad((f1 * r.to(DEV)).sum(), i1)
        monkeypatch.delenv("W2E_VIT_V1")
        assert not torch.equal(f1, f2), "both runs took the same kernels"
        assert_close(f2, f1, 1e-4, "features v2 vs v1"), assert_close(g2, g1, 1e-3, "image gradient v2 vs v1")
        if b == 4:
            io = img.clone().requires_grad_(True)
            fo = OC.encode_image(sd, io)
            (go,) = torch.autograd.grad((fo * r).sum(), io)
            assert_close(f2, fo, 1e-3, "features vs oracle"), assert_close(g2, go, 2e-3, "image gradient vs oracle")


def _unpack(p, m):
    """[K/4, mpad, 4] K-quad-major -> [m, K] row-major."""
    return p[:, :m].permute(1, 0, 2).reshape(m, -1)


def test_reduce_gelu_and_the_packed_outputs_of_the_tower_kernels():
    """w2e_reduce_gelu (the QuickGELU pair / its derivative on a sum of split-K slabs) against float64, and every producer's
    `packed_rows` form -- w2e_reduce_ln_fwd, w2e_layernorm_bwd_part, w2e_reduce_gelu (both modes), w2e_attn2_fwd / _bwd -- against its own
    row-major output: the K-quad-major tensor (w2e_gemm_pk's A operand) must hold the same values, bit for bit."""
    from where2edit_amd import vit_hip as V
    from where2edit_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(9)
    for m in (200, 8):
        mpad, n = -(-m // 32) * 32, 3072
        part = torch.randn(3, m, n, generator=g).to(DEV)
        bias, aux = torch.randn(n, generator=g).to(DEV), torch.randn(m, n, generator=g).to(DEV)
        h, act = torch.empty(m, n, device=DEV), torch.empty(m, n, device=DEV)
        call("w2e_reduce_gelu", ptr(part), 3, m * n, ptr(bias), None, ptr(h), ptr(act), m, n, 0, 0, stream_ptr())
        href = part.double().sum(0) + bias.double()
        assert_close(h, href, 1e-6, "h"), assert_close(act, href * torch.sigmoid(1.702 * href), 1e-5, "QuickGELU(h)")
        h2, actp = torch.empty(m, n, device=DEV), torch.empty(n // 4, mpad, 4, device=DEV)
        call("w2e_reduce_gelu", ptr(part), 3, m * n, ptr(bias), None, ptr(h2), ptr(actp), m, n, 0, mpad, stream_ptr())
        assert torch.equal(h2, h) and torch.equal(_unpack(actp, m), act), "mode 0 packed"
        out = torch.empty(m, n, device=DEV)
        call("w2e_reduce_gelu", ptr(part), 3, m * n, None, ptr(aux), ptr(out), None, m, n, 1, 0, stream_ptr())
        s_ = torch.sigmoid(1.702 * aux.double())
        assert_close(out, part.double().sum(0) * (s_ * (1 + 1.702 * aux.double() * (1 - s_))), 1e-5, "gelu-grad")
        outp = torch.empty(n // 4, mpad, 4, device=DEV)
        call("w2e_reduce_gelu", ptr(part), 3, m * n, None, ptr(aux), ptr(outp), None, m, n, 1, mpad, stream_ptr())
        assert torch.equal(_unpack(outp, m), out), "mode 1 packed"
    m, dim, mpad = 150, 768, 160
    part = torch.randn(2, m, dim, generator=g).to(DEV)
    gamma, beta = (torch.randn(dim, generator=g) * 0.2 + 1).to(DEV), (torch.randn(dim, generator=g) * 0.2).to(DEV)
    x, y, mean, rstd = V._reduce_ln(part, None, None, gamma, beta, 1e-5)
    _, yp, _, _ = V._reduce_ln(part, None, None, gamma, beta, 1e-5, mpad=mpad)
    assert yp.shape == (dim // 4, mpad, 4) and torch.equal(_unpack(yp, m), y), "LayerNorm packed"
    gx = V._ln_bwd_part(part, x, gamma, mean, rstd, None)
    gx2, gxp = V._ln_bwd_part(part, x, gamma, mean, rstd, None, mpad=mpad)
    assert torch.equal(gx2, gx) and torch.equal(_unpack(gxp, m), gx), "LayerNorm backward packed"
    B, L, H = 3, 50, 12
    d3, mm = 3 * H * 64, 3 * 50
    slabs, bias = torch.randn(2, mm, d3, generator=g).to(DEV), torch.randn(d3, generator=g).to(DEV)
    o, op = torch.empty(mm, H * 64, device=DEV), torch.empty(H * 16, mpad, 4, device=DEV)
    call("w2e_attn2_fwd", ptr(slabs), 2, mm * d3, ptr(bias), ptr(o), B, L, H, 0, stream_ptr())
    call("w2e_attn2_fwd", ptr(slabs), 2, mm * d3, ptr(bias), ptr(op), B, L, H, mpad, stream_ptr())
    assert torch.equal(_unpack(op, mm), o), "attention packed"
    gs = torch.randn(1, mm, H * 64, generator=g).to(DEV)
    gq, gqp = torch.empty(mm, d3, device=DEV), torch.empty(d3 // 4, mpad, 4, device=DEV)
    call("w2e_attn2_bwd", ptr(slabs), 2, mm * d3, ptr(bias), ptr(gs), 1, mm * H * 64, ptr(gq), B, L, H, 0, stream_ptr())
    call("w2e_attn2_bwd", ptr(slabs), 2, mm * d3, ptr(bias), ptr(gs), 1, mm * H * 64, ptr(gqp), B, L, H, mpad, stream_ptr())
    assert torch.equal(_unpack(gqp, mm), gq), "attention backward packed"


@pytest.mark.parametrize("b,t,d", [(4, 1, 512), (3, 5, 200), (8, 2, 768)])
def test_fused_loss_tail_equals_the_stock_composition(b, t, d):
    """csrc/losstail.hip: (1) the tail of CLIP.forward / CLIPLoss.forward -- normalise, exp(logit_scale), cosine logits, 1 - ./100 -- and
    (2) coach.calc_loss's clip_lambda * mean + l2_lambda * MSE as one launch each way, against the stock-op composition in float64
    (clip_loss.py:16, coach.py:223-245): values, the gradient to the image features and to w_hat."""
    from where2edit_amd import vit_hip as V
    g = torch.Generator().manual_seed(b * 100 + t * 10 + d)
    feat = torch.randn(b, d, generator=g).to(DEV).requires_grad_(True)
    text = torch.randn(t, d, generator=g).to(DEV)
    ls = torch.tensor(2.6593, device=DEV)
    w, w_hat = torch.randn(b, 18, 512, generator=g).to(DEV), torch.randn(b, 18, 512, generator=g).to(DEV).requires_grad_(True)
    for sim_flag in (False, True):
        out = V.clip_logits(feat, text, ls, similarity=sim_flag)
        fd = feat.detach().double().requires_grad_(True)
        ref = ls.double().exp() * (fd / fd.norm(dim=1, keepdim=True)) @ (text.double() / text.double().norm(dim=1, keepdim=True)).t()
        ref = 1 - ref / 100 if sim_flag else ref
        assert_close(out, ref, 1e-5, "logits" if not sim_flag else "similarity")
        r = torch.randn(b, t, generator=g).to(DEV)
        (gf,) = torch.autograd.grad((out * r).sum(), feat)
        (gr,) = torch.autograd.grad((ref * r.double()).sum(), fd)
        assert_close(gf, gr, 1e-5, "d logits / d features")
    sim = V.clip_logits(feat, text, ls, similarity=True)
    loss, l_clip, l_l2 = V.step_loss(sim, w_hat, w, 1.0, 0.8)
    assert not l_clip.requires_grad and not l_l2.requires_grad and loss.requires_grad
    wd = w_hat.detach().double().requires_grad_(True)
    fd = feat.detach().double().requires_grad_(True)
    sim_ref = 1 - ls.double().exp() * (fd / fd.norm(dim=1, keepdim=True)) @ (text.double() / text.double().norm(dim=1, keepdim=True)).t() / 100
    ref_clip, ref_l2 = sim_ref.mean(), torch.nn.functional.mse_loss(wd, w.double())
    ref = ref_clip * 1.0 + ref_l2 * 0.8
    assert_close(loss, ref, 1e-5, "loss"), assert_close(l_clip, ref_clip, 1e-5, "loss_clip"), assert_close(l_l2, ref_l2, 1e-5, "loss_l2")
    loss.backward()
    ref.backward()
    assert_close(w_hat.grad, wd.grad, 1e-5, "d loss / d w_hat")
    assert_close(feat.grad, fd.grad, 1e-5, "d loss / d features (through both fused nodes)")


@pytest.mark.parametrize("m,n,k", [(200, 2304, 768), (200, 768, 3072), (400, 3072, 768), (50, 768, 768), (77, 200, 104)])
def test_gemm_pk_and_its_packing_vs_float64(m, n, k):
    """csrc/vit3.hip: w2e_pack_kq (plain and transposed) and w2e_gemm_pk (both operands K-quad-major, register-fed, one wave per tile and
    K slice, split-K slabs) against float64: the tower's shapes at batch 4 / 8 / 1, and a ragged one (rows and columns that pad, a K whose
    last slice is short); every admissible split count incl. the one w2e_gemm_pk_splits picks."""
    from where2edit_amd import _lib
    from where2edit_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).to(DEV)
    w = torch.randn(n, k, generator=g).to(DEV)
    mpad, npad = -(-m // 32) * 32, -(-n // 64) * 64
    ap = torch.full((k // 4, mpad, 4), float("nan"), device=DEV)
    wp = torch.empty((k // 4, npad, 4), device=DEV)
    call("w2e_pack_kq", ptr(a), ptr(ap), m, mpad, k, k, 0, stream_ptr())
    call("w2e_pack_kq", ptr(w), ptr(wp), n, npad, k, k, 0, stream_ptr())
    assert torch.equal(ap[:, :m].permute(1, 0, 2).reshape(m, k), a) and float(ap[:, m:].abs().sum()) == 0.0
    wt = w.t().contiguous()  # [k, n]: the transposed packing must give the same operand
    wp2 = torch.empty_like(wp)
    call("w2e_pack_kq", ptr(wt), ptr(wp2), n, npad, k, n, 1, stream_ptr())
    assert torch.equal(wp2, wp)
    ref = a.double() @ w.double().t()
    pick = _lib.load().w2e_gemm_pk_splits(m, n, k)
    chunks = k // 8
    tried = 0
    for sp in sorted({1, 2, 3, 5, 8, pick}):
        per = -(-(-(-chunks // sp)) // 4) * 4
        if sp > chunks or (sp - 1) * per >= chunks:
            continue
        c = torch.full((sp, m, n), float("nan"), device=DEV)
        call("w2e_gemm_pk", ptr(ap), ptr(wp), ptr(c), m, n, k, mpad, npad, n, sp, stream_ptr())
        assert_close(c.double().sum(0), ref, 5e-6, f"gemm_pk, {sp} slabs")  # (fp32 sums over up to 3072 terms)
        tried += 1
    assert tried >= 2


@pytest.mark.parametrize("m,n,k", [(200, 2304, 768), (200, 768, 3072), (400, 3072, 768), (50, 768, 768), (77, 200, 112)])
def test_gemm_pk_h_is_the_fp16_rounding_of_its_operands(m, n, k):
    """The opt-in fp16-operand GEMM (w2e_pack_kq_h + w2e_gemm_pk_h): exactly A and W rounded to fp16 (nearest even), products and sums in
    fp32 -- compared with a float64 product of the ROUNDED operands at fp32 accumulation accuracy, plain and transposed weight packs,
    every admissible split count; and against the unrounded product to show what the rounding costs (printed by -s)."""
    from where2edit_amd import _lib
    from where2edit_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(3 * m + n + k)
    a = torch.randn(m, k, generator=g).to(DEV)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(DEV)
    mpad, npad = -(-m // 32) * 32, -(-n // 64) * 64
    ap = torch.empty((k // 4, mpad, 4), device=DEV)
    call("w2e_pack_kq", ptr(a), ptr(ap), m, mpad, k, k, 0, stream_ptr())
    wh = torch.empty((k // 8, npad, 8), device=DEV, dtype=torch.float16)
    call("w2e_pack_kq_h", ptr(w), wh.data_ptr(), n, npad, k, k, 0, stream_ptr())
    wh2 = torch.empty_like(wh)
    call("w2e_pack_kq_h", ptr(w.t().contiguous()), wh2.data_ptr(), n, npad, k, n, 1, stream_ptr())
    assert torch.equal(wh, wh2), "plain and transposed fp16 packs differ"
    # the pack holds fp16(W) at k = 16 s + 8 (c >> 2) + 4 h + (c & 3) for entry (s, h), component c
    wr = wh[:, :n].float().reshape(k // 16, 2, n, 2, 4).permute(2, 0, 3, 1, 4).reshape(n, k)
    assert torch.equal(wr, w.half().float()), "fp16 pack is not round-to-nearest of W in the documented k order"
    ref_h = a.half().double() @ w.half().double().t()
    ref = a.double() @ w.double().t()
    steps = k // 16
    pick = _lib.load().w2e_gemm_pk_h_splits(m, n, k)
    tried = 0
    for sp in sorted({1, 2, 3, pick}):
        per = -(-(-(-steps // sp)) // 4) * 4
        if sp > steps or (sp - 1) * per >= steps:
            continue
        c = torch.full((sp, m, n), float("nan"), device=DEV)
        call("w2e_gemm_pk_h", ptr(ap), wh.data_ptr(), ptr(c), m, n, k, mpad, npad, n, sp, stream_ptr())
        assert_close(c.double().sum(0), ref_h, 5e-6, f"gemm_pk_h, {sp} slabs, against the rounded operands")
        tried += 1
    assert tried >= 2
    print(f"fp16 operands, K = {k}: |C_h - C| / |C| = {rel_err(c.double().sum(0), ref):.2e}")
    with pytest.raises(RuntimeError, match="K %% 16|K % 16"):
        call("w2e_gemm_pk_h", ptr(ap), wh.data_ptr(), ptr(c), m, n, 104, mpad, npad, n, 1, stream_ptr())


def test_vit_b32_tower_with_fp16_operands_against_the_fp32_tower_and_the_oracle():
    """CLIP.set_precision("f16") (opt-in; what the reference's GPU tower computes: fp16 Linear operands, fp32 accumulation here): the
    image features and the image gradient of ViT-B/32 against the default fp32 tower and against the fp32 CPU oracle, with the measured
    errors printed; the four block GEMMs run on w2e_gemm_pk_h, everything else is the fp32 path; switching back restores the fp32
    results bit for bit in deterministic terms (same kernels, same packs)."""
    cfg = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch=32, context_length=8,
               vocab_size=64, text_width=64, text_layers=1)
    m, sd = _model(cfg)
    img = seeded.tensor("clip.f16.img", (4, 3, 224, 224), 0.5)
    r = seeded.tensor("clip.f16.r", (4, 512))

    def run():
        ig = img.to(DEV).requires_grad_(True)
        f = m.encode_image(ig)
        (g,) = torch.autograd.grad((f * r.to(DEV)).sum(), ig)
        return f.detach(), g

    f32, g32 = run()
    m.set_precision("f16")
    try:
        f16, g16 = run()
    finally:
        m.set_precision("f32")
    f32b, g32b = run()
    assert_close(f32b, f32, 1e-6, "fp32 tower after switching back"), assert_close(g32b, g32, 1e-5, "fp32 gradient after switching back")
    assert not torch.equal(f16, f32), "the fp16 switch changed nothing"
    io = img.clone().requires_grad_(True)
    fo = OC.encode_image(sd, io)
    (go,) = torch.autograd.grad((fo * r).sum(), io)
    e_f, e_g = rel_err(f16, f32), rel_err(g16, g32)
    cos = torch.nn.functional.cosine_similarity(g16.flatten().double(), g32.flatten().double(), dim=0).item()
    print(f"fp16-operand tower vs fp32 tower: features {e_f:.2e}, image gradient {e_g:.2e} (cosine {cos:.6f}); "
          f"vs the fp32 oracle: features {rel_err(f16, fo):.2e}, gradient {rel_err(g16, go):.2e}")
    # fp16 has an 11-bit significand (2^-11 = 4.9e-4 per operand); measured on MI355X (profiles/r05_clip_f16.txt): features 4.7e-4,
    # image gradient 9.3e-4, cosine 1.000000 -- held to twice / three times that
    assert e_f <= 1e-3 and e_g <= 3e-3 and cos >= 0.99999, (e_f, e_g, cos)
    assert_close(f16, fo, 1e-3, "fp16-operand features vs the fp32 oracle")
