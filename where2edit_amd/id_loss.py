"""criteria/id_loss.py + models/facial_recognition surface: `IDLoss(opts)(y_hat, y) -> (loss, 0)` on the IR-SE50
ArcFace backbone, with the reference's module tree so `model_ir_se50.pth` loads unchanged.

SURVEY 8(a) A9 / 8(f) N1: on the GPU, in eval mode with frozen weights (the only way the reference uses it), the
backbone runs on the hand-written kernels of irse_hip.py / include/w2e_irse.h: the 3x3 convolutions on the fp32-MFMA
engine of the StyleGAN2 layers with BatchNorm folded and PReLU fused, the SE block on its own kernels.  The module tree
below is the reference's (state_dict keys) and doubles as the CPU / train-mode execution on stock ops."""
from collections import namedtuple

import torch
from torch import nn
from torch.nn import (AdaptiveAvgPool2d, BatchNorm1d, BatchNorm2d, Conv2d, Dropout, Linear, MaxPool2d, Module, PReLU,
                      ReLU, Sequential, Sigmoid)


class Flatten(Module):
    def forward(self, input):
        return input.view(input.size(0), -1)


def l2_norm(input, axis=1):
    return input / torch.norm(input, 2, axis, True)


Bottleneck = namedtuple("Block", ["in_channel", "depth", "stride"])


def get_block(in_channel, depth, num_units, stride=2):
    return [Bottleneck(in_channel, depth, stride)] + [Bottleneck(depth, depth, 1) for _ in range(num_units - 1)]


def get_blocks(num_layers):
    """models/facial_recognition/helpers.py:29-53"""
    units = {50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}
    if num_layers not in units:
        raise ValueError("Invalid number of layers: {}. Must be one of [50, 100, 152]".format(num_layers))
    u = units[num_layers]
    return [get_block(64, 64, u[0]), get_block(64, 128, u[1]), get_block(128, 256, u[2]), get_block(256, 512, u[3])]


class SEModule(Module):
    """helpers.py:56-72"""

    def __init__(self, channels, reduction):
        super().__init__()
        self.avg_pool = AdaptiveAvgPool2d(1)
        self.fc1 = Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        self.relu = ReLU(inplace=True)
        self.fc2 = Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)
        self.sigmoid = Sigmoid()

    def forward(self, x):
        return x * self.sigmoid(self.fc2(self.relu(self.fc1(self.avg_pool(x)))))


class bottleneck_IR(Module):
    """helpers.py:75-94"""

    def __init__(self, in_channel, depth, stride, se=False):
        super().__init__()
        if in_channel == depth:
            self.shortcut_layer = MaxPool2d(1, stride)
        else:
            self.shortcut_layer = Sequential(Conv2d(in_channel, depth, (1, 1), stride, bias=False), BatchNorm2d(depth))
        layers = [BatchNorm2d(in_channel), Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False), PReLU(depth),
                  Conv2d(depth, depth, (3, 3), stride, 1, bias=False), BatchNorm2d(depth)]
        if se:
            layers.append(SEModule(depth, 16))
        self.res_layer = Sequential(*layers)

    def forward(self, x):
        return self.res_layer(x) + self.shortcut_layer(x)


class bottleneck_IR_SE(bottleneck_IR):
    """helpers.py:97-119"""

    def __init__(self, in_channel, depth, stride):
        super().__init__(in_channel, depth, stride, se=True)


class Backbone(Module):
    """models/facial_recognition/model_irse.py:9-48"""

    def __init__(self, input_size, num_layers, mode="ir", drop_ratio=0.4, affine=True):
        super().__init__()
        assert input_size in [112, 224], "input_size should be 112 or 224"
        assert num_layers in [50, 100, 152], "num_layers should be 50, 100 or 152"
        assert mode in ["ir", "ir_se"], "mode should be ir or ir_se"
        unit = bottleneck_IR if mode == "ir" else bottleneck_IR_SE
        self.input_layer = Sequential(Conv2d(3, 64, (3, 3), 1, 1, bias=False), BatchNorm2d(64), PReLU(64))
        side = 7 if input_size == 112 else 14
        self.output_layer = Sequential(BatchNorm2d(512), Dropout(drop_ratio), Flatten(), Linear(512 * side * side, 512),
                                       BatchNorm1d(512, affine=affine))
        self.body = Sequential(*[unit(b.in_channel, b.depth, b.stride) for blk in get_blocks(num_layers) for b in blk])

    def forward(self, x, n_grad=None):
        """GPU + eval mode + frozen weights (the way criteria/id_loss.py uses it): the hand-written kernels
        (irse_hip.backbone_forward).  Otherwise -- CPU tensors, train mode, trainable weights -- the stock module tree."""
        if x.is_cuda and not self.training and x.dtype == torch.float32 and not any(p.requires_grad for p in self.parameters()):
            from . import irse_hip
            key = tuple((p.data_ptr(), p._version) for p in self.parameters())
            if getattr(self, "_plan_key", None) != key:
                self._plan = irse_hip.BackbonePlan(self)
                self._plan_key = key
            return irse_hip.backbone_forward(self._plan, x, n_grad)
        if x.is_cuda and not getattr(self, "_warned_stock", False):
            import warnings
            warnings.warn("IR-SE50 Backbone is running on stock PyTorch ops (train mode or trainable weights): the HIP path needs "
                          ".eval() and requires_grad_(False)")
            self._warned_stock = True
        return l2_norm(self.output_layer(self.body(self.input_layer(x))))


class IDLoss(nn.Module):
    """criteria/id_loss.py:7-40.  `opts.ir_se50_weights` (a state_dict file) is optional here: without it the
    backbone keeps its random init (synthetic benchmarking; there is no network to fetch the pretrained file)."""

    def __init__(self, opts, gpu=0):
        super().__init__()
        self.facenet = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se")
        path = getattr(opts, "ir_se50_weights", None)
        if path is not None:
            self.facenet.load_state_dict(torch.load(path, map_location="cpu"))
        self.pool = torch.nn.AdaptiveAvgPool2d((256, 256))
        self.face_pool = torch.nn.AdaptiveAvgPool2d((112, 112))
        self.facenet.eval()
        for p in self.facenet.parameters():
            p.requires_grad_(False)
        self.opts = opts

    def extract_feats(self, x):
        if x.is_cuda and x.shape[2] == x.shape[3] and x.shape[2] % 256 == 0 and x.dtype == torch.float32:
            from . import functional as K  # fused pool -> crop -> pool (K5b); other sizes / CPU: the literal chain below
            return self.facenet(K.id_preprocess(x))
        if x.shape[2] != 256:
            x = self.pool(x)
        x = x[:, :, 35:223, 32:220]  # crop interesting region (id_loss.py:22)
        return self.facenet(self.face_pool(x))

    def forward(self, y_hat, y):
        if y_hat.is_cuda and y_hat.shape == y.shape and y_hat.shape[2] == y_hat.shape[3] and y_hat.shape[2] % 256 == 0 \
                and not self.facenet.training:
            # one pass over [y_hat; y]: twice the rows per launch for the launch-bound 14^2 / 7^2 stages; only the y_hat half
            # takes part in the backward (y's features are detached in the reference, id_loss.py:33)
            from . import functional as K
            n = y_hat.shape[0]
            faces = torch.cat([K.id_preprocess(y_hat), K.id_preprocess(y.detach())])  # (cat the 112^2 crops, not the images)
            feats = self.facenet(faces, n_grad=n)
            y_hat_feats, y_feats = feats[:n], feats[n:].detach()
        else:
            y_feats = self.extract_feats(y).detach()
            y_hat_feats = self.extract_feats(y_hat)
        loss = (1 - (y_hat_feats * y_feats).sum(1)).mean()  # mean_i (1 - <f(y_hat_i), f(y_i)>)  (id_loss.py:34-40)
        return loss, 0
