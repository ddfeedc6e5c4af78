"""criteria/clip_loss.py surface: `CLIPLoss(opts)(image, text) -> 1 - logits/100`, with `.model`,
`.upsample`, `.avg_pool` attributes (callers poke them: run_attention.py:1008,1126,1163-1164).

Differences by design: the 7x nearest up-sample + 32x32 average pool runs as ONE closed-form HIP kernel
(the reference materialises a [B,3,7168,7168] tensor, 617 MB per image); the ViT runs in fp32 on the
hand-written kernels; text features are cached.  Weights: `opts.clip_weights` (a state_dict in OpenAI
key layout) when given, otherwise random-init ViT-B/32 -- there is no network to download the
pretrained model the reference fetches at construction (clip_loss.py:10)."""
import torch

from . import functional as K
from .clip_vit import CLIP


class CLIPLoss(torch.nn.Module):
    def __init__(self, opts, model=None):
        super().__init__()
        if model is None:
            model = CLIP()
            path = getattr(opts, "clip_weights", None)
            if path is not None:
                sd = torch.load(path, map_location="cpu")
                sd = sd.get("state_dict", sd)
                sd = {k: v.float() for k, v in sd.items() if k not in ("input_resolution", "context_length", "vocab_size")}
                model.load_state_dict(sd, strict=True)
        self.model = model.float().eval()
        for p in self.model.parameters():
            p.requires_grad_(False)  # CLIP is a fixed critic on this path
        self.upsample = torch.nn.Upsample(scale_factor=7)
        self.avg_pool = torch.nn.AvgPool2d(kernel_size=opts.stylegan_size // 32)
        self.stylegan_size = opts.stylegan_size

    def preprocess(self, image):
        """avg_pool(upsample(image)) without the 49x intermediate (clip_loss.py:15)."""
        if image.shape[-1] != self.stylegan_size or image.shape[-2] != self.stylegan_size:
            if not getattr(self, "_warned_literal", False):
                import warnings
                warnings.warn(f"CLIPLoss.preprocess: image {tuple(image.shape[-2:])} is not stylegan_size {self.stylegan_size}: running the "
                              "literal Upsample(7) -> AvgPool chain on stock ops (it materialises the 49x image)")
                self._warned_literal = True
            return self.avg_pool(self.upsample(image))  # sizes the closed form does not cover: literal chain
        return K.clip_preprocess(image)

    def forward(self, image, text):
        image = self.preprocess(image)
        if hasattr(self.model, "logits_per_image"):  # the package's CLIP: normalise, scale, product and 1 - ./100 in one launch
            return self.model.logits_per_image(image, text, similarity=True)
        similarity = 1 - self.model(image, text)[0] / 100
        return similarity
