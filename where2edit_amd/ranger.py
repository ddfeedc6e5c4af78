"""mapper/training/ranger.py surface: `Ranger(params, lr, alpha, k, N_sma_threshhold, betas, eps,
weight_decay, use_gc, gc_conv_only)` = RAdam + Lookahead + gradient centralisation, same update rule
and the same per-parameter state keys (`step`, `exp_avg`, `exp_avg_sq`, `slow_buffer`), so optimizer
state round-trips.  Host logic on [512,512]-sized tensors: multi-tensor (`torch._foreach_*`) updates
instead of the reference's Python loop of ~10 tiny ops per parameter."""
import math

import torch
from torch.optim.optimizer import Optimizer


class Ranger(Optimizer):
    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-5,
                 weight_decay=0, use_gc=True, gc_conv_only=False):
        if not 0.0 <= alpha <= 1.0:
            raise ValueError(f"Invalid slow update rate: {alpha}")
        if not 1 <= k:
            raise ValueError(f"Invalid lookahead steps: {k}")
        if not lr > 0:
            raise ValueError(f"Invalid Learning Rate: {lr}")
        if not eps > 0:
            raise ValueError(f"Invalid eps: {eps}")
        defaults = dict(lr=lr, alpha=alpha, k=k, step_counter=0, betas=betas, N_sma_threshhold=N_sma_threshhold,
                        eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.N_sma_threshhold = N_sma_threshhold
        self.alpha = alpha
        self.k = k
        self.use_gc = use_gc
        self.gc_gradient_threshold = 3 if gc_conv_only else 1

    @staticmethod
    def _radam_step_size(step, beta1, beta2, threshold):
        """ranger.py:124-137: length of the approximated SMA and the rectified step size."""
        beta2_t = beta2 ** step
        n_max = 2 / (1 - beta2) - 1
        n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
        if n_sma > threshold:
            size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) \
                / (1 - beta1 ** step)
        else:
            size = 1.0 / (1 - beta1 ** step)
        return n_sma, size

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Ranger optimizer does not support sparse gradients")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, dtype=torch.float32)
                    state["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32)
                    state["slow_buffer"] = p.detach().clone()
                state["step"] += 1
                by_step.setdefault(state["step"], []).append(p)
            for step, params in by_step.items():
                grads = [p.grad.float() for p in params]
                if self.use_gc:  # gradient centralisation (ranger.py:112): g -= mean over all dims but the first
                    by_shape = {}
                    for i, g in enumerate(grads):
                        if g.dim() > self.gc_gradient_threshold:
                            by_shape.setdefault(tuple(g.shape), []).append(i)
                    for idx in by_shape.values():  # same-shaped gradients (the mapper's 12 [512,512] weights): one stacked mean
                        st = torch.stack([grads[i] for i in idx])  # a copy: p.grad itself stays as autograd left it
                        st.sub_(st.mean(dim=tuple(range(2, st.dim())), keepdim=True))
                        for i, g in zip(idx, st.unbind(0)):
                            grads[i] = g
                exp_avg = [self.state[p]["exp_avg"] for p in params]
                exp_avg_sq = [self.state[p]["exp_avg_sq"] for p in params]
                torch._foreach_mul_(exp_avg_sq, beta2)
                torch._foreach_addcmul_(exp_avg_sq, grads, grads, value=1 - beta2)
                torch._foreach_mul_(exp_avg, beta1)
                torch._foreach_add_(exp_avg, grads, alpha=1 - beta1)
                n_sma, step_size = self._radam_step_size(step, beta1, beta2, self.N_sma_threshhold)
                if group["weight_decay"] != 0:
                    torch._foreach_mul_(params, 1 - group["weight_decay"] * group["lr"])
                if n_sma > self.N_sma_threshhold:
                    denom = torch._foreach_sqrt(exp_avg_sq)
                    torch._foreach_add_(denom, group["eps"])
                    torch._foreach_addcdiv_(params, exp_avg, denom, value=-step_size * group["lr"])
                else:
                    torch._foreach_add_(params, exp_avg, alpha=-step_size * group["lr"])
                if step % group["k"] == 0:  # lookahead sync (ranger.py:158-161)
                    slow = [self.state[p]["slow_buffer"] for p in params]
                    diff = torch._foreach_sub(params, slow)
                    torch._foreach_add_(slow, diff, alpha=self.alpha)
                    torch._foreach_copy_(params, slow)
        return loss
