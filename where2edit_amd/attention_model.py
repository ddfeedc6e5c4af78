"""attention/attention_model.py surface: the Generator whose forward additionally (a) records every
post-layer activation (`return_features`) and (b), given `attention_map`, alpha-blends layer
`attention_layer` -- and the ToRGB output that follows it -- with the cached features of the unedited
image: out = m*out + (1-m)*feature_map[layer-1], m = nearest-resized mask (attention_model.py:473-676).
The building blocks are the ones of where2edit_amd.stylegan2 (the reference file duplicates them);
the blend is one HIP kernel (K6) instead of interpolate + repeat + 3 elementwise passes."""
from . import functional as K
from .stylegan2 import (Blur, ConstantInput, Downsample, EqualConv2d, EqualLinear, ModulatedConv2d,  # noqa: F401
                        NoiseInjection, PixelNorm, ScaledLeakyReLU, StyledConv, ToRGB, Upsample, make_kernel)
from .stylegan2 import Generator as _BaseGenerator
from .op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d  # noqa: F401


class Generator(_BaseGenerator):
    def forward(self, styles, return_latents=False, return_features=False, inject_index=None, truncation=1,
                truncation_latent=None, input_is_latent=False, input_is_stylespace=False, noise=None,
                randomize_noise=True, attention_layer=0, attention_map=None, feature_map=None):
        latent, noise = self._prepare(styles, inject_index, truncation, truncation_latent, input_is_latent,
                                      input_is_stylespace, noise, randomize_noise)
        recorded = []
        state = {"armed": False}  # `this_layer` of attention_model.py:532

        def on_layer(n, is_rgb, act):
            if attention_map is not None:
                layer = n + 1
                if layer == attention_layer or (is_rgb and state["armed"]):
                    state["armed"] = not is_rgb
                    act = K.mask_blend(act, feature_map[layer - 1], attention_map)
            recorded.append(act)
            return act

        hooks = None
        if not return_features:  # nothing to record: only the blended layer and the ToRGB after it need the hook; every other
            hooks = set()        # layer keeps the fused training forms of the plain generator
            if attention_map is not None and attention_layer > 0:
                plan = self._layers()
                n0 = attention_layer - 1
                hooks.add(n0)
                hooks.update([n for n in range(n0 + 1, len(plan)) if plan[n][1]][:1])
        image, style_vector = self._synthesis(latent, noise, input_is_stylespace, on_layer, hooks)
        if return_latents:
            return image, latent, style_vector
        if return_features:
            return image, latent, style_vector, recorded
        return image, None
