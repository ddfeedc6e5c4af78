"""Generates tests/golden/e4e.npz by running the REFERENCE's Encoder4Editing and GradualStyleEncoder
(models/encoders/psp_encoders.py:58-200) in this container on seeded weights (tests/golden/seeded.py: irse_fill keyed by
the reference's own state_dict names, loaded strict=True).  Never runs on the GPU box.

    python tests/golden/make_golden_e4e.py
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


def encoder_state_dict(state_dict):
    """irse_fill, with the style blocks' conv biases / latlayer biases near 0 and every conv ~ fan_in^-0.5."""
    return seeded.irse_fill(state_dict, salt=7)


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from models.encoders import psp_encoders as P
    opts = types.SimpleNamespace(stylegan_size=1024)
    x = seeded.tensor("e4e.x", (1, 3, 256, 256), 0.5)
    store = {}
    for name, cls in (("e4e", P.Encoder4Editing), ("gse", P.GradualStyleEncoder)):
        torch.manual_seed(0)
        net = cls(50, "ir_se", opts).eval()
        sd = encoder_state_dict(net.state_dict())
        net.load_state_dict(sd, strict=True)
        with torch.no_grad():
            w = net(x)
        store[name + ".w"] = w.numpy()
        store[name + ".keys"] = np.asarray(sorted(sd))
        store[name + ".shapes"] = np.asarray([str(tuple(sd[k].shape)) for k in sorted(sd)])
        print(name, w.shape, float(w.abs().mean()))
        if name == "e4e":  # the FPN taps (psp_encoders.py:176-183)
            with torch.no_grad():
                y = net.input_layer(x)
                for i, l in enumerate(net.body):
                    y = l(y)
                    if i in (6, 20, 23):
                        store[f"e4e.c{(6, 20, 23).index(i) + 1}_strided"] = y[:, ::8, ::4, ::4].numpy()
    np.savez_compressed(os.path.join(HERE, "e4e.npz"), **store)


if __name__ == "__main__":
    main()
