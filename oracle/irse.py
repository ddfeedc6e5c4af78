"""Oracle (CPU, test-only): the ArcFace IR-SE50 identity embedder and loss, functional over the reference's
state_dict (models/facial_recognition/model_irse.py:9-48, helpers.py:29-119, criteria/id_loss.py:19-40).
Eval mode throughout (the reference calls facenet.eval(), id_loss.py:14): BatchNorm uses its running statistics
and Dropout is the identity.  Pinned against the reference's own Backbone by tests/golden/irse.npz."""
import torch
import torch.nn.functional as F

UNITS_50 = ((64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3))  # helpers.py:29-36: (in_channel, depth, num_units)


def blocks(units=UNITS_50):
    """[(in_channel, depth, stride)] in body order (helpers.py:24-26: first unit of a stage has stride 2)."""
    out = []
    for cin, depth, n in units:
        out.append((cin, depth, 2))
        out += [(depth, depth, 1)] * (n - 1)
    return out


def _bn(sd, p, x, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def _unit(sd, p, x, cin, depth, stride):
    """bottleneck_IR_SE (helpers.py:97-119): BN -> conv3x3 -> PReLU -> conv3x3(stride) -> BN -> SE, plus the shortcut
    (MaxPool2d(1, stride) = strided subsampling when cin == depth, else conv1x1(stride) + BN)."""
    if cin == depth:
        sc = x[:, :, ::stride, ::stride]
    else:
        sc = _bn(sd, p + ".shortcut_layer.1", F.conv2d(x, sd[p + ".shortcut_layer.0.weight"], stride=stride))
    r = _bn(sd, p + ".res_layer.0", x)
    r = F.conv2d(r, sd[p + ".res_layer.1.weight"], padding=1)
    r = F.prelu(r, sd[p + ".res_layer.2.weight"])
    r = F.conv2d(r, sd[p + ".res_layer.3.weight"], stride=stride, padding=1)
    r = _bn(sd, p + ".res_layer.4", r)
    g = r.mean((2, 3), keepdim=True)  # SEModule (helpers.py:56-72)
    g = torch.sigmoid(F.conv2d(F.relu(F.conv2d(g, sd[p + ".res_layer.5.fc1.weight"])), sd[p + ".res_layer.5.fc2.weight"]))
    return r * g + sc


def backbone(sd, x, return_stages=False):
    """Backbone(112, 50, 'ir_se').forward (model_irse.py:44-48): [B,3,112,112] -> L2-normalised [B,512]."""
    x = F.prelu(_bn(sd, "input_layer.1", F.conv2d(x, sd["input_layer.0.weight"], padding=1)), sd["input_layer.2.weight"])
    stages = []
    for i, (cin, depth, stride) in enumerate(blocks()):
        x = _unit(sd, f"body.{i}", x, cin, depth, stride)
        if return_stages:
            stages.append(x)
    x = _bn(sd, "output_layer.0", x).flatten(1)
    x = F.linear(x, sd["output_layer.3.weight"], sd["output_layer.3.bias"])
    x = F.batch_norm(x, sd["output_layer.4.running_mean"], sd["output_layer.4.running_var"], sd["output_layer.4.weight"],
                     sd["output_layer.4.bias"], False, 0.0, 1e-5)
    x = x / torch.norm(x, 2, 1, True)
    return (x, stages) if return_stages else x


def extract_feats(sd, img):
    """id_loss.py:19-25: pool to 256^2 (unless already), crop [35:223, 32:220], pool to 112^2, embed."""
    if img.shape[2] != 256:
        img = F.adaptive_avg_pool2d(img, (256, 256))
    return backbone(sd, F.adaptive_avg_pool2d(img[:, :, 35:223, 32:220], (112, 112)))


def id_loss(sd, y_hat, y):
    """id_loss.py:27-40: mean_i (1 - <f(y_hat_i), f(y_i).detach()>)."""
    fy = extract_feats(sd, y).detach()
    fh = extract_feats(sd, y_hat)
    return (1 - (fh * fy).sum(1)).mean()
