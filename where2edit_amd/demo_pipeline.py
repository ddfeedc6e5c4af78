"""BASELINE configs[4], the inference pipeline of show_demo/try_demo.py:93-157 + utils_demo.py:142-157 as one function:

    256^2 image -> e4e (+ latent_avg)               -> W+ [B,18,512]                     (try_demo.py:98)
    G(W+) with return_latents                       -> the 26 S-space codes              (:99-101; here Generator.style_codes:
                                                       the modulation affines only, no synthesis pass for a dropped image)
    G(S codes) with return_features                 -> img_orig + 26 activations (+ const input)   (:122-125)
    CLIP image features of img_orig                                                           (:123-124)
    Mapper(text (+) S codes, features)              -> new S codes + region mask          (utils_demo.py:152)
    mask[mask < attention_threshold] = 0, gaussian_blur(5)                                (:154-155)
    G(new codes, blend at attention_layer under the mask)  -> edited image                (:156)
    CLIP image features of the edited image                                               (try_demo.py:148-149)

No backward anywhere; every rank of an N-GPU run processes its own images (replicas only)."""
import torch
import torch.nn.functional as F

from .run_attention import info_nce  # noqa: F401  (re-export for callers that score edits)


def gaussian_blur5(x):
    """torchvision.transforms.functional.gaussian_blur(x, 5) on a [B,1,s,s] mask (sigma 1.1, reflect padding) -- [B,64,64]-sized."""
    t = torch.linspace(-2.0, 2.0, steps=5, device=x.device)
    pdf = torch.exp(-0.5 * (t / 1.1).pow(2))
    k1 = pdf / pdf.sum()
    k2 = (k1[:, None] * k1[None, :]).to(x.dtype)
    c = x.shape[1]
    return F.conv2d(F.pad(x, [2, 2, 2, 2], mode="reflect"), k2.expand(c, 1, 5, 5), groups=c)


@torch.no_grad()
def invert_and_edit(images, e4e, g_ema, clip_loss, mapper, text_features, attention_text_features, *, attention_layer=13,
                    strength_alpha=0.1, attention_threshold=0.8):
    """images [B,3,256,256] in [-1,1]; text_features / attention_text_features [B,512] (CLIP text embeddings; the
    reference tokenises with OpenAI's BPE, which is not in this image).  Returns a dict with img_orig, img_gen, the
    mask, the inverted W+ and the new S codes, and the CLIP image features before / after."""
    b = images.shape[0]
    latents = e4e(images)
    if hasattr(g_ema, "style_codes"):  # the codes alone: the reference runs the whole generator here and drops the image
        latents, styles = g_ema.style_codes([latents], input_is_latent=True)
    else:
        _, latents, styles = g_ema([latents], input_is_latent=True, return_latents=True, randomize_noise=False)
    img_orig, _, _, feats = g_ema([styles], input_is_latent=True, randomize_noise=False, return_features=True, input_is_stylespace=True)
    feats = list(feats) + [g_ema.input.input.repeat(b, 1, 1, 1)]
    feat_orig = clip_loss.model.encode_image(clip_loss.preprocess(img_orig))
    blend_size = feats[attention_layer - 1].shape[-1]
    x = [torch.cat([text_features.unsqueeze(1), s[:, :, :, 0, 0]], dim=-1) for s in styles]
    new_codes, _, _ = mapper(x, feats, blend_size, attention_text=attention_text_features, strength_alpha=strength_alpha)
    # The demo's own copy of the net returns the RAW cluster-pooled map (show_demo/utils_demo.py:135-139: the threshold and
    # blur lines of the training forward are commented out there) and one_text_edit thresholds and blurs it ONCE
    # (utils_demo.py:154-155).  The training net's forward returns the thresholded + blurred map; thresholding that again
    # would zero the blurred edges -- so the raw map is taken from the net's last forward, not from its return value.
    mask = mapper.last["same"].unsqueeze(1)
    mask = gaussian_blur5(torch.where(mask < attention_threshold, torch.zeros_like(mask), mask))
    img_gen, _, _, _ = g_ema([new_codes], input_is_latent=True, randomize_noise=False, return_features=True, input_is_stylespace=True,
                             attention_layer=attention_layer, attention_map=mask, feature_map=feats)
    feat_gen = clip_loss.model.encode_image(clip_loss.preprocess(img_gen))
    return {"img_orig": img_orig, "img_gen": img_gen, "mask": mask, "latents": latents, "new_codes": new_codes,
            "features_orig": feat_orig, "features_gen": feat_gen}


def capture_invert_and_edit(images, e4e, g_ema, clip_loss, mapper, text_features, attention_text_features, *, warmup=3, **kw):
    """The fixed-shape pipeline as ONE hipGraph: returns `run(images, text_features, attention_text_features) -> dict` that copies
    the inputs into static buffers, replays the graph and returns the (static) output tensors.  About 1500 launches per call are
    otherwise enqueued eagerly and the pipeline is host-bound (40 ms of kernels in 52 ms of wall time at batch 8).  As for
    Coach.capture_step, a pipeline that issues memset operations is refused (they are not replayed reliably on this ROCm stack)."""
    static = [images.clone(), text_features.clone(), attention_text_features.clone()]

    def body():
        return invert_and_edit(static[0], e4e, g_ema, clip_loss, mapper, static[1], static[2], **kw)

    from .coach import capture_graph  # (warm-up, memset check, thread-local capture: the same helper as Coach.capture_step)
    graph, out = capture_graph(body, "capture_invert_and_edit: the pipeline", images.device, warmup)

    def run(images_new, text_new, attention_text_new):
        static[0].copy_(images_new), static[1].copy_(text_new), static[2].copy_(attention_text_new)
        graph.replay()
        return out

    run.graph = graph
    return run

