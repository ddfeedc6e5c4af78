"""N1: the IR-SE50 identity-loss network on the hand-written kernels (include/w2e_irse.h) -- the generic conv entry
point against torch convolutions in float64, the backbone against the fixture captured from the reference's Backbone
(tests/golden/irse.npz) and against the oracle (oracle/irse.py) including the input gradient, and IDLoss as a whole."""
import types

import pytest
import torch
import torch.nn.functional as F

import seeded
from helpers import assert_close, assert_grad_close, golden
from oracle import irse as OI

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("b,k,n,h,w", [(2, 64, 64, 28, 28), (1, 3, 64, 112, 112), (2, 24, 40, 14, 18), (3, 128, 128, 7, 7)])
def test_conv3x3_same_with_folded_bn_and_prelu(b, k, n, h, w):
    from where2edit_amd import functional as K
    from where2edit_amd import irse_hip as I
    g = torch.Generator().manual_seed(k * 7 + n)
    wt = torch.randn(n, k, 3, 3, generator=g) * (k * 9) ** -0.5
    x = torch.randn(b, k, h, w, generator=g)
    a, bias, slope = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g), torch.rand(n, generator=g) * 0.5 + 0.05
    ref = F.prelu(F.conv2d(x.double(), wt.double(), padding=1) * a.double()[None, :, None, None] + bias.double()[None, :, None, None],
                  slope.double())
    pack = K.conv_pack(wt.to(DEV), 1.0, False, False)
    y = I.conv3x3(x.to(DEV), pack, n, h, w, out_scale=a.to(DEV)[None].repeat(b, 1).contiguous(), bias=bias.to(DEV), slope=slope.to(DEV))
    assert_close(y, ref, 1e-4, "conv + BN + PReLU")
    y = I.conv3x3(x.to(DEV), pack, n, h, w, bias=bias.to(DEV))
    assert_close(y, F.conv2d(x.double(), wt.double(), padding=1) + bias.double()[None, :, None, None], 1e-4, "conv + bias")


@pytest.mark.parametrize("b,k,n,h", [(2, 64, 64, 28), (1, 40, 72, 9), (2, 256, 512, 7)])
def test_conv3x3_stride2_pad1_and_its_adjoint(b, k, n, h):
    """DOWN with down_pad = nn.Conv2d(k, n, 3, 2, 1) on a [b,k,2h,2h] input; UP + the (+1,+1) crop = its input gradient."""
    from where2edit_amd import functional as K
    from where2edit_amd import irse_hip as I
    g = torch.Generator().manual_seed(k + n + h)
    wt = torch.randn(n, k, 3, 3, generator=g) * (k * 9) ** -0.5
    x = torch.randn(b, k, 2 * h, 2 * h, generator=g)
    a, bias = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    xd = x.double().requires_grad_(True)
    ref = F.conv2d(xd, wt.double(), stride=2, padding=1) * a.double()[None, :, None, None] + bias.double()[None, :, None, None]
    y = I.conv3x3(x.to(DEV), K.conv_pack(wt.to(DEV), 1.0, False, False), n, h, h, mode=K.MODE_DOWN, down_pad=1,
                  out_scale=a.to(DEV)[None].repeat(b, 1).contiguous(), bias=bias.to(DEV))
    assert_close(y, ref, 1e-4, "stride-2 pad-1 conv + BN")
    gy = torch.randn(b, n, h, h, generator=g)
    (gref,) = torch.autograd.grad(ref, xd, gy.double())
    tt = I.conv3x3(gy.to(DEV), K.conv_pack(wt.to(DEV), 1.0, True, False), k, h, h, mode=K.MODE_UP, in_scale=a.to(DEV)[None].repeat(b, 1).contiguous())
    gx = I.affine_act_bwd(tt, None, None, None, b, k, 2 * h, 2 * h, planar=True)
    assert_close(gx, gref, 1e-4, "input gradient of the stride-2 conv")
    # centre-tap pack = the 1x1 stride-2 shortcut convolution (helpers.py:103-106)
    w1 = torch.randn(n, k, 1, 1, generator=g) * k ** -0.5
    w9 = torch.zeros(n, k, 3, 3)
    w9[:, :, 1, 1] = w1[:, :, 0, 0]
    y = I.conv3x3(x.to(DEV), K.conv_pack(w9.to(DEV), 1.0, False, False), n, h, h, mode=K.MODE_DOWN, down_pad=1)
    assert_close(y, F.conv2d(x.double(), w1.double(), stride=2), 1e-4, "1x1 stride-2 shortcut")


def _backbone():
    from where2edit_amd.id_loss import Backbone
    net = Backbone(112, 50, drop_ratio=0.6, mode="ir_se").eval()
    sd = seeded.irse_fill(net.state_dict())
    net.load_state_dict(sd, strict=True)
    return net.to(DEV).requires_grad_(False), sd


def test_irse50_backbone_on_hip_matches_reference_fixture_and_oracle_gradient():
    net, sd = _backbone()
    x = seeded.tensor("irse.x", (2, 3, 112, 112), 0.5)
    xg = x.to(DEV).requires_grad_(True)
    y = net(xg)
    assert hasattr(net, "_plan"), "the HIP path did not run"
    assert_close(y, golden("irse")["feats"], 1e-4, "IR-SE50 features vs the reference's Backbone")
    r = seeded.tensor("irse.r", (2, 512))
    (gx,) = torch.autograd.grad((y * r.to(DEV)).sum(), xg)
    xo = x.clone().requires_grad_(True)
    yo = OI.backbone(sd, xo)
    (go,) = torch.autograd.grad((yo * r).sum(), xo)
    assert_grad_close(gx, go, "IR-SE50 input gradient")


def test_id_loss_on_hip_matches_oracle_with_gradient():
    """criteria/id_loss.py:19-40 end to end at 256^2 and 1024^2-shaped inputs: the fused pool-crop-pool, the batched
    [y_hat; y] embedding pass, loss value and d loss / d y_hat."""
    from where2edit_amd.id_loss import IDLoss
    mod = IDLoss(types.SimpleNamespace(ir_se50_weights=None))
    sd = seeded.irse_fill(mod.facenet.state_dict())
    mod.facenet.load_state_dict(sd, strict=True)
    mod = mod.to(DEV)
    for size in (256, 512):
        y = seeded.tensor(f"id.y{size}", (2, 3, size, size), 0.5)
        yh = (y + 0.2 * seeded.tensor(f"id.d{size}", (2, 3, size, size))).clone()
        yg = yh.to(DEV).requires_grad_(True)
        loss, zero = mod(yg, y.to(DEV))
        (g,) = torch.autograd.grad(loss, yg)
        yo = yh.clone().requires_grad_(True)
        lo = OI.id_loss(sd, yo, y)
        (go,) = torch.autograd.grad(lo, yo)
        assert zero == 0 and abs(float(loss) - float(lo)) <= 1e-4 * max(abs(float(lo)), 1e-3)
        assert_grad_close(g, go, f"d id_loss / d y_hat at {size}")
