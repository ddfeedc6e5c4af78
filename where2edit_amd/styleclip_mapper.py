"""mapper/styleclip_mapper.py surface: `StyleCLIPMapper(opts)` with `.mapper`, `.decoder`, `.face_pool`,
`.opts`, the same checkpoint handling (`get_keys`, strict flags) and the same forward, including the
in-place latent_mask / inject_latent / alpha editing of the codes (styleclip_mapper.py:14-77)."""
import torch
from torch import nn

from . import latent_mappers
from .stylegan2 import Generator, freeze_conv_weights


def get_keys(d, name):
    """styleclip_mapper.py:7-11"""
    if "state_dict" in d:
        d = d["state_dict"]
    return {k[len(name) + 1:]: v for k, v in d.items() if k[:len(name)] == name}


class StyleCLIPMapper(nn.Module):
    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        self.mapper = self.set_mapper()
        self.decoder = Generator(self.opts.stylegan_size, 512, 8)
        self.face_pool = torch.nn.AdaptiveAvgPool2d((256, 256))
        self.load_weights()
        # The reference leaves requires_grad=True on the decoder and never optimises it (coach.py:174-180: net.mapper only), so
        # its conv-weight gradients are computed and thrown away.  The HIP conv kernels compute none and refuse a weight that asks
        # for one (stylegan2._trainable_weight) -- so a wrapper built the reference's way (parameters trainable by default, only
        # the mapper handed to the optimizer) is made to work by freezing exactly those weights here; everything else of the
        # decoder (biases, noise strengths, modulation affines, the constant input) keeps its flag and its gradient.
        freeze_conv_weights(self.decoder)

    def set_mapper(self):
        if self.opts.work_in_stylespace:
            return latent_mappers.WithoutToRGBStyleSpaceMapper(self.opts)
        if self.opts.mapper_type == "SingleMapper":
            return latent_mappers.SingleMapper(self.opts)
        if self.opts.mapper_type == "LevelsMapper":
            return latent_mappers.LevelsMapper(self.opts)
        raise Exception("{} is not a valid mapper".format(self.opts.mapper_type))

    def load_weights(self):
        if getattr(self.opts, "checkpoint_path", None) is not None:
            print("Loading from checkpoint: {}".format(self.opts.checkpoint_path))
            ckpt = torch.load(self.opts.checkpoint_path, map_location="cpu")
            self.mapper.load_state_dict(get_keys(ckpt, "mapper"), strict=True)
            self.decoder.load_state_dict(get_keys(ckpt, "decoder"), strict=True)
        elif getattr(self.opts, "stylegan_weights", None) is not None:
            print("Loading decoder weights from pretrained!")
            ckpt = torch.load(self.opts.stylegan_weights, map_location="cpu")
            self.decoder.load_state_dict(ckpt["g_ema"], strict=False)
        # else: synthetic / externally loaded weights (bench, tests) -- the reference requires a file here

    def forward(self, x, resize=True, latent_mask=None, input_code=False, randomize_noise=True, inject_latent=None,
                return_latents=False, alpha=None):
        codes = x if input_code else self.mapper(x)
        if latent_mask is not None:
            for i in latent_mask:
                if inject_latent is not None:
                    if alpha is not None:
                        codes[:, i] = alpha * inject_latent[:, i] + (1 - alpha) * codes[:, i]
                    else:
                        codes[:, i] = inject_latent[:, i]
                else:
                    codes[:, i] = 0
        input_is_latent = not input_code
        # the reference unpacks two values here, which raises with return_latents=True because the decoder
        # then returns a 3-tuple (Q10); the first two entries are what that code meant
        res = self.decoder([codes], input_is_latent=input_is_latent, randomize_noise=randomize_noise,
                           return_latents=return_latents)
        images, result_latent = res[0], res[1]
        if resize:
            images = self.face_pool(images)
        if return_latents:
            return images, result_latent
        return images
