"""Oracle (CPU, test-only): one StyleCLIP-mapper training step, the unit of the
headline metric (mapper/training/coach.py:70-92 + calc_loss :223-245), and the
Ranger update (mapper/training/ranger.py:78-164)."""
import math

import torch
import torch.nn.functional as F

from . import clip_model, mappers, stylegan2


def mapper_step_loss(gen_sd, mapper_sd, clip_sd, w, tokens, *, size, clip_lambda=1.0, latent_l2_lambda=0.8,
                     id_lambda=0.0, id_feats=None, mapper="levels"):
    """x = G(w) (no grad); w_hat = w + 0.1*M(w); x_hat = G(w_hat);
    L = id_lambda*L_id + clip_lambda*mean(CLIPLoss(x_hat, t)) + l2_lambda*MSE(w_hat, w)
    (coach.py:81-90, 223-245).  `id_feats(img)->[B,512]` is the ArcFace embedder
    when id_lambda > 0 (criteria/id_loss.py:19-40)."""
    with torch.no_grad():
        x, _ = stylegan2.generator_forward(gen_sd, [w], size=size, input_is_latent=True, randomize_noise=False)
    m = mappers.levels_mapper(mapper_sd, w) if mapper == "levels" else mappers.single_mapper(mapper_sd, w)
    w_hat = w + 0.1 * m
    x_hat, w_hat, _ = stylegan2.generator_forward(gen_sd, [w_hat], size=size, input_is_latent=True,
                                                  randomize_noise=False, return_latents=True)
    terms = {}
    loss = 0.0
    if id_lambda > 0:
        fy = id_feats(x).detach()
        fh = id_feats(x_hat)
        l_id = (1 - (fh * fy).sum(1)).mean()  # id_loss.py:34-40: mean_i (1 - <f_hat_i, f_i>)
        terms["loss_id"] = l_id
        loss = loss + id_lambda * l_id
    if clip_lambda > 0:
        l_clip = clip_model.clip_loss(clip_sd, x_hat, tokens, size).mean()
        terms["loss_clip"] = l_clip
        loss = loss + clip_lambda * l_clip
    if latent_l2_lambda > 0:
        l_l2 = F.mse_loss(w_hat, w)
        terms["loss_l2_latent"] = l_l2
        loss = loss + latent_l2_lambda * l_l2
    terms["loss"] = loss
    return loss, terms, x, x_hat, w_hat


class RangerState:
    """Per-parameter state of mapper/training/ranger.py (RAdam + lookahead + gradient
    centralisation), restated functionally.  Defaults are the reference's."""

    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, n_sma_threshold=5, betas=(0.95, 0.999), eps=1e-5):
        self.lr, self.alpha, self.k, self.thr, self.betas, self.eps = lr, alpha, k, n_sma_threshold, betas, eps
        self.step_n = 0
        self.exp_avg = [torch.zeros_like(p) for p in params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in params]
        self.slow = [p.detach().clone() for p in params]

    @torch.no_grad()
    def step(self, params, grads):
        b1, b2 = self.betas
        self.step_n += 1
        t = self.step_n
        b2t = b2 ** t
        n_max = 2 / (1 - b2) - 1
        n_sma = n_max - 2 * t * b2t / (1 - b2t)
        if n_sma > self.thr:  # ranger.py:131-137
            step_size = math.sqrt((1 - b2t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) \
                / (1 - b1 ** t)
        else:
            step_size = 1.0 / (1 - b1 ** t)
        for i, (p, g) in enumerate(zip(params, grads)):
            g = g.clone()
            if g.dim() > 1:  # gradient centralisation, conv+fc (ranger.py:112-113)
                g -= g.mean(dim=tuple(range(1, g.dim())), keepdim=True)
            self.exp_avg_sq[i].mul_(b2).addcmul_(g, g, value=1 - b2)
            self.exp_avg[i].mul_(b1).add_(g, alpha=1 - b1)
            if n_sma > self.thr:
                p.addcdiv_(self.exp_avg[i], self.exp_avg_sq[i].sqrt().add_(self.eps), value=-step_size * self.lr)
            else:
                p.add_(self.exp_avg[i], alpha=-step_size * self.lr)
            if t % self.k == 0:  # lookahead (ranger.py:158-161)
                self.slow[i].add_(p - self.slow[i], alpha=self.alpha)
                p.copy_(self.slow[i])
