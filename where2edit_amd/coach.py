"""mapper/training/coach.py surface, reduced to the hot path: `Coach(opts)` builds the same objects
(net = StyleCLIPMapper, clip_loss, id_loss, latent_l2_loss, optimizer over net.mapper.parameters() only)
and `train_step(w)` is one iteration of the reference's `train()` loop body (coach.py:79-92):

    x     = G(w)                       (no grad)
    w_hat = w + 0.1 * M(w)
    x_hat = G(w_hat)
    L     = id_lambda*L_id(x_hat,x) + clip_lambda*mean(CLIPLoss(x_hat,text)) + l2_lambda*MSE(w_hat,w)
    L.backward(); [all-reduce mapper grads]; optimizer.step()

TensorBoard / image grids / dataset plumbing are out of scope; `train()` iterates a latent
tensor with the same shuffle/drop_last semantics and `checkpoint_me()` writes the reference's schema."""
import os

import torch
from torch import nn

from . import dist as w2e_dist
from .clip_loss import CLIPLoss
from .ranger import Ranger
from .styleclip_mapper import StyleCLIPMapper


FUSED_LOSS_TAIL = os.environ.get("W2E_FUSED_LOSS_TAIL", "1") != "0"  # (A/B aid: "0" keeps the stock-op composition of calc_loss)


class Coach:
    def __init__(self, opts, net=None, clip_loss=None, id_loss=None, text_inputs=None, device=None, data_parallel=False):
        self.opts = opts
        self.global_step = 0
        self.device = device if device is not None else "cuda:0"  # coach.py:25
        self.opts.device = self.device
        self.net = (net if net is not None else StyleCLIPMapper(self.opts)).to(self.device)
        # The decoder is never optimised (configure_optimizers: mapper parameters only, coach.py:174-180).  The
        # reference nevertheless leaves requires_grad=True on it and pays for ~30 M unused weight gradients per
        # step; here it is frozen, which changes no result the loop can observe.
        self.net.decoder.requires_grad_(False)
        if self.opts.id_lambda > 0:
            if id_loss is None:
                from .id_loss import IDLoss
                id_loss = IDLoss(self.opts)
            self.id_loss = id_loss.to(self.device).eval()
        if self.opts.clip_lambda > 0:
            self.clip_loss = (clip_loss if clip_loss is not None else CLIPLoss(opts)).to(self.device)
        if self.opts.latent_l2_lambda > 0:
            self.latent_l2_loss = nn.MSELoss().to(self.device).eval()
        self.optimizer = self.configure_optimizers()
        # text tokens: the reference tokenises opts.description with OpenAI's BPE (coach.py:55); the tokenizer
        # is not in this image, so callers pass token ids ([n_text, 77] int64) or get a fixed synthetic prompt
        if text_inputs is None:
            text_inputs = synthetic_tokens(1)
        self.text_inputs = text_inputs.to(self.device)
        self.bucket = w2e_dist.GradBucket(self.net.mapper.parameters()) if data_parallel else None
        self.best_val_loss = None
        # x = G(w) (no grad) does not depend on the mapper.  With W2E_SIDE_STREAM=1 it is enqueued on a second HIP
        # stream so that its large conv launches fill the CUs that the launch-bound parts of the main stream (ViT at
        # M = 50*B, the 4^2..32^2 layers, [B,512] style math) leave idle: 0-5 % more images/s depending on the device.
        # Off by default: overlapped kernels stretch each other's durations, which blurs per-kernel roofline numbers.
        self._side = torch.cuda.Stream(device=self.device) if os.environ.get("W2E_SIDE_STREAM", "0") == "1" else None
        self._side_primed = False
        # one merged generator pass for x and x_hat (forward_pair); W2E_MERGE_FORWARD=0 keeps the reference's two passes
        self.merge_forward = os.environ.get("W2E_MERGE_FORWARD", "1") != "0"

    def configure_optimizers(self):
        params = list(self.net.mapper.parameters())  # mapper only: the decoder is never optimised (coach.py:174-180)
        if self.opts.optim_name == "adam":
            return torch.optim.Adam(params, lr=self.opts.learning_rate)
        return Ranger(params, lr=self.opts.learning_rate)

    def forward_pair(self, w, mask=None):
        """coach.py:80-89 (W+ and S-space branches).  With `mask` ([B,1,s,s], or a callable features -> mask) and opts.attention_layer > 0 (BASELINE
        configs[2]) the edited image is generated the region-attention way (attention/run_attention.py:1104-1126): the
        decoder is attention_model.Generator, the unedited pass also returns its 26 activations, and the edited pass
        blends layer `attention_layer` (and the ToRGB after it) with them under the mask."""
        dec = self.net.decoder
        s_space = getattr(self.opts, "work_in_stylespace", False)
        att_layer = getattr(self.opts, "attention_layer", 0)
        if mask is not None and att_layer > 0:
            with torch.no_grad():
                x, _, _, feats = dec([w], input_is_latent=True, randomize_noise=False, truncation=1, return_features=True,
                                     input_is_stylespace=s_space)
            self._x_ready = None
            if callable(mask):  # the region-attention net's mask branch, fed with the unedited pass's activations
                mask = mask(feats)  # (run_attention.py:1231-1245: the mask is a function of the cached features)
            if s_space:  # the S-space blend sites of attention_model.py:573-588, 637-660 (run_attention.py:1245)
                delta = self.net.mapper(w)
                w_hat = [c + 0.1 * dc for c, dc in zip(w, delta)]
                x_hat, _, w_hat = dec([w_hat], input_is_latent=True, return_latents=True, randomize_noise=False, truncation=1,
                                      input_is_stylespace=True, attention_layer=att_layer, attention_map=mask, feature_map=feats)
                return x, x_hat, w_hat
            w_hat = w + 0.1 * self.net.mapper(w)
            x_hat, w_hat, _ = dec([w_hat], input_is_latent=True, return_latents=True, randomize_noise=False, truncation=1,
                                  attention_layer=att_layer, attention_map=mask, feature_map=feats)
            return x, x_hat, w_hat
        if self.merge_forward and self._side is None and s_space and torch.is_grad_enabled():
            # the same merged pass for S-space codes (coach.py:84-89 with work_in_stylespace): every code tensor is [c; c_hat]
            from . import functional as K
            n = w[0].shape[0]
            delta = self.net.mapper(w)
            w_hat = [c + 0.1 * dc for c, dc in zip(w, delta)]
            with K.nograd_prefix(n):
                both, _, codes = dec([[torch.cat([c.detach(), ch]) for c, ch in zip(w, w_hat)]], input_is_latent=True, return_latents=True,
                                     randomize_noise=False, truncation=1, input_is_stylespace=True)
            self._x_ready = None
            return both[:n].detach(), K.tail_rows(both, n), [c[n:] for c in codes]
        if self.merge_forward and self._side is None and not s_space and torch.is_grad_enabled():
            # x = G(w) (no grad) and x_hat = G(w_hat) as ONE generator pass over [w; w_hat]: twice the rows per launch (the
            # same kernels run ~9 % faster per image at twice the batch) and half the forward launches.  Values are the ones of
            # the two separate passes (per-sample arithmetic); the backward of every generator node works on the w_hat rows
            # only (functional.nograd_prefix), so no gradient work is spent on the no-grad half.
            from . import functional as K
            n = w.shape[0]
            w_hat = w + 0.1 * self.net.mapper(w)
            with K.nograd_prefix(n):
                both, lat, _ = dec([torch.cat([w.detach(), w_hat])], input_is_latent=True, return_latents=True, randomize_noise=False,
                                   truncation=1)
            self._x_ready = None
            return both[:n].detach(), K.tail_rows(both, n), lat[n:]
        main = torch.cuda.current_stream()
        if self._side is not None and not self._side_primed:
            # the decoder's derived-weight caches (conv packs, wsq, stacked affines) are built lazily by the first pass:
            # build them on the MAIN stream once, so the side-stream pass and G(w_hat) never race on half-written packs
            with torch.no_grad():
                dec([w[:1]] if not s_space else [[c[:1] for c in w]], input_is_latent=True, randomize_noise=False, truncation=1,
                    input_is_stylespace=s_space)
            self._side_primed = True
        if self._side is not None:
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side), torch.no_grad():
                x, _ = dec([w], input_is_latent=True, randomize_noise=False, truncation=1, input_is_stylespace=s_space)
            x.record_stream(main)
            self._x_ready = self._side.record_event()
        else:
            with torch.no_grad():
                x, _ = dec([w], input_is_latent=True, randomize_noise=False, truncation=1, input_is_stylespace=s_space)
            self._x_ready = None
        if s_space:
            delta = self.net.mapper(w)
            w_hat = [c + 0.1 * dc for c, dc in zip(w, delta)]
            x_hat, _, w_hat = dec([w_hat], input_is_latent=True, return_latents=True, randomize_noise=False,
                                  truncation=1, input_is_stylespace=True)
        else:
            w_hat = w + 0.1 * self.net.mapper(w)
            x_hat, w_hat, _ = dec([w_hat], input_is_latent=True, return_latents=True, randomize_noise=False, truncation=1)
        return x, x_hat, w_hat

    def calc_loss(self, w, x, w_hat, x_hat):
        """coach.py:223-245; the loss dict holds 0-dim tensors (no host sync inside the step)."""
        loss_dict = {}
        loss = 0.0
        if getattr(self, "_x_ready", None) is not None:  # x was produced on the side stream
            torch.cuda.current_stream().wait_event(self._x_ready)
            self._x_ready = None
        if self.opts.id_lambda > 0:
            loss_id, sim_improvement = self.id_loss(x_hat, x)
            loss_dict["loss_id"] = loss_id.detach()
            loss_dict["id_improve"] = sim_improvement
            loss = loss_id * self.opts.id_lambda
        s_space = getattr(self.opts, "work_in_stylespace", False)
        if self.opts.clip_lambda > 0 and self.opts.latent_l2_lambda > 0 and not s_space and torch.is_tensor(w_hat) and w_hat.is_cuda \
                and w_hat.dtype == torch.float32 and torch.is_tensor(w) and w.shape == w_hat.shape and w.dtype == w_hat.dtype \
                and w.device == w_hat.device and FUSED_LOSS_TAIL:  # (any other pairing: the stock composition below, as the reference)
            # the clip and latent-L2 terms and their weighted sum as ONE launch (vit_hip.step_loss; backward: one more) instead of the
            # mean / mse_loss / mul / add chain and its autograd mirror
            from . import vit_hip
            sim = self.clip_loss(x_hat, self.text_inputs)
            tail, loss_dict["loss_clip"], loss_dict["loss_l2_latent"] = vit_hip.step_loss(sim, w_hat, w, self.opts.clip_lambda,
                                                                                            self.opts.latent_l2_lambda)
            loss = tail if self.opts.id_lambda <= 0 else loss + tail
            loss_dict["loss"] = loss.detach()
            return loss, loss_dict
        if self.opts.clip_lambda > 0:
            loss_clip = self.clip_loss(x_hat, self.text_inputs).mean()
            loss_dict["loss_clip"] = loss_clip.detach()
            loss = loss + loss_clip * self.opts.clip_lambda
        if self.opts.latent_l2_lambda > 0:
            if getattr(self.opts, "work_in_stylespace", False):
                loss_l2_latent = sum(self.latent_l2_loss(c_hat, c) for c_hat, c in zip(w_hat, w))
            else:
                loss_l2_latent = self.latent_l2_loss(w_hat, w)
            loss_dict["loss_l2_latent"] = loss_l2_latent.detach()
            loss = loss + loss_l2_latent * self.opts.latent_l2_lambda
        loss_dict["loss"] = loss.detach()
        return loss, loss_dict

    def train_step(self, w, mask=None):
        """One mapper step on this rank's shard of latents (the unit of the headline metric)."""
        if self.bucket is not None:
            self.bucket.zero()
        else:
            self.optimizer.zero_grad()
        x, x_hat, w_hat = self.forward_pair(w, mask)
        loss, loss_dict = self.calc_loss(w, x, w_hat, x_hat)
        loss.backward()
        if self.bucket is not None:
            self.bucket.all_reduce_mean()
        self.optimizer.step()
        self.global_step += 1
        return loss_dict

    def accumulated_step(self, chunks, step=None):
        """One mapper step on a shard that is processed as `len(chunks)` EQUAL micro-batches (bench.py --scaling strong: global batch
        64 fixed, so one GPU's shard of 64 / N latents runs as micro-batches of 8 -- a single pass over 64 latents would put the
        32 @ 1024^2 activations of the merged [w; w_hat] pass past the 4 GB one buffer descriptor covers and 32-bit element counts).
        The loss terms are per-sample means (coach.py:223-245), so the gradient of the shard's mean loss is the mean of the
        micro-batches' gradients: they are summed with weight 1/len(chunks) in an fp32 accumulator the size of the bucket, written
        back, then ONE all-reduce and ONE optimizer step -- the same update as the unsplit step up to summation order.
        `step`: a function from Coach.capture_step captured on a tensor of a chunk's shape (replayed with finish=False); None: eager.
        Needs the flat gradient bucket (Coach(..., data_parallel=True); at world size 1 its all-reduce is skipped)."""
        if self.bucket is None:
            raise RuntimeError("accumulated_step needs the flat gradient bucket: build the Coach with data_parallel=True")
        n = len(chunks)
        if n == 1:
            return step(chunks[0]) if step is not None else self.train_step(chunks[0])
        if getattr(self, "_acc", None) is None:
            self._acc = torch.zeros_like(self.bucket.flat)
        self._acc.zero_()
        sums = {}
        for c in chunks:
            if c.shape != chunks[0].shape:
                raise ValueError("accumulated_step: micro-batches must have equal shapes (the mean of means is the mean only then)")
            if step is not None:
                d = step(c, finish=False)
            else:
                self.bucket.zero()
                x, x_hat, w_hat = self.forward_pair(c)
                loss, d = self.calc_loss(c, x, w_hat, x_hat)
                loss.backward()
            self._acc.add_(self.bucket.flat, alpha=1.0 / n)
            for k_, v in d.items():  # (a graphed step returns the SAME static tensors every replay: fold them in now)
                if torch.is_tensor(v):
                    sums[k_] = v.detach() / n if k_ not in sums else sums[k_] + v.detach() / n
        self.bucket.flat.copy_(self._acc)
        self.bucket.all_reduce_mean()
        self.optimizer.step()
        self.global_step += 1
        return sums

    # ---- the fixed-shape step as ONE hipGraph launch -------------------------------------------------------------
    def capture_step(self, w, mask=None, warmup=3, side_stream=False):
        """Capture zero-grad + forward_pair + calc_loss + backward for inputs of w's shape into a hipGraph (about 330
        kernel / memset nodes at 1024^2) and return `step(w[, mask]) -> loss_dict`, which copies the inputs into the
        graph's static buffers, replays it, then runs the gradient all-reduce and the optimizer eagerly (Ranger's
        rectification and look-ahead are host-side control flow, ranger.py:124-161).  The libw2e.so entry points are
        capture-safe (stream-ordered, no allocation, no synchronisation; include/w2e.h); `warmup` eager iterations on the
        capture stream first build every lazily cached pack and opt the large-LDS kernels in on this device."""
        from . import profiling
        if profiling._active is not None:
            raise RuntimeError("capture_step: per-kernel HIP-event timing cannot be recorded inside a graph")
        # `side_stream`: inside the graph the no-grad G(w) pass forks onto a second stream (its large conv launches fill the CUs
        # that the launch-bound parts of the main branch leave idle: +1-2 % images/s); eager steps keep one stream, so that
        # per-kernel HIP-event durations are not stretched by overlap
        eager_side = self._side
        try:
            if side_stream and self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            s_space = getattr(self.opts, "work_in_stylespace", False)
            static_w = [c.clone() for c in w] if s_space else w.clone()
            # a callable mask (features -> mask, e.g. the region-attention net's mask branch) is captured WITH the step: it must be
            # stream-ordered like everything else (no host synchronisation, no .item(); the warm-up iterations below fill its caches)
            static_mask = mask if (mask is None or callable(mask)) else mask.clone()
            params = list(self.net.mapper.parameters())

            def body():
                if self.bucket is not None:
                    self.bucket.zero()
                else:
                    for p in params:  # no zero fill + accumulate per parameter: the backward's own gradient tensors (static
                        p.grad = None  # addresses in the graph's memory pool) become the .grad of every replay
                x, x_hat, w_hat = self.forward_pair(static_w, static_mask)
                loss, loss_dict = self.calc_loss(static_w, x, w_hat, x_hat)
                loss.backward()
                return loss_dict

            graph, static_out = capture_graph(body, "capture_step: the step", self.device, warmup, leaves=params)
        finally:
            self._side = eager_side
        static_grads = [p.grad for p in params]

        def step(w_new, mask_new=None, finish=True):
            """`finish=False`: replay only -- the gradients are left in the mapper's .grad (the bucket), no all-reduce, no optimizer
            step (Coach.accumulated_step sums several such replays before it finishes the step once)."""
            if s_space:
                for dst, src in zip(static_w, w_new):
                    dst.copy_(src)
            else:
                static_w.copy_(w_new)
            if static_mask is not None and mask_new is not None and not callable(static_mask):
                static_mask.copy_(mask_new)
            graph.replay()
            if self.bucket is None:  # an eager step in between (optimizer.zero_grad()) may have detached them
                for p, g in zip(params, static_grads):
                    p.grad = g
            if not finish:
                return static_out
            if self.bucket is not None:
                self.bucket.all_reduce_mean()
            self.optimizer.step()
            self.global_step += 1
            return static_out

        step.graph = graph
        return step

    def train(self, latents, max_steps=None, generator=None):
        """Epochs over a [N,18,512] latent tensor: shuffle, batch_size, drop_last (coach.py:44-48,70-79)."""
        self.net.train()
        max_steps = max_steps if max_steps is not None else self.opts.max_steps
        bs = self.opts.batch_size
        log = []
        while self.global_step < max_steps:
            perm = torch.randperm(latents.shape[0], generator=generator)
            for i in range(0, latents.shape[0] - bs + 1, bs):
                if self.global_step >= max_steps:
                    break
                log.append(self.train_step(latents[perm[i:i + bs]].to(self.device)))
        return log

    @torch.no_grad()
    def validate(self, latents):
        """coach.py:122-161 without image logging: mean loss dict over held-out latents."""
        self.net.eval()
        agg = []
        bs = getattr(self.opts, "test_batch_size", 1)
        for i in range(0, min(latents.shape[0], 201 * bs) - bs + 1, bs):
            w = latents[i:i + bs].to(self.device)
            x, x_hat, w_hat = self.forward_pair(w)
            _, d = self.calc_loss(w, x, w_hat, x_hat)
            agg.append({k: float(v) for k, v in d.items()})
        self.net.train()
        return {k: sum(d[k] for d in agg) / len(agg) for k in agg[0]} if agg else {}

    def checkpoint_me(self, path):
        """coach.py:163-172,267-272: {'state_dict': net.state_dict(), 'opts': vars(opts)}."""
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save({"state_dict": self.net.state_dict(), "opts": {k: v for k, v in vars(self.opts).items()}}, path)


def capture_graph(body, what, device, warmup=3, leaves=()):
    """The one way this package turns a fixed-shape, stream-ordered `body()` into a hipGraph (Coach.capture_step and
    demo_pipeline.capture_invert_and_edit both come here): `warmup` eager runs on a side stream (they build every lazily cached
    pack and opt the large-LDS kernels in on this device), the memset check below, then the capture.  Returns (graph, body's
    return value inside the capture = the static outputs).
    capture_error_mode="thread_local": with a process group alive, RCCL's watchdog thread queries events on its own; under the
    default "global" mode any such call from another thread invalidates a capture in progress (seen with bench.py --gpus N;
    `bench.py --workload 5 --gpus N` keeps a group alive around the pipeline's capture in the same way)."""
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream())
    # `leaves` (the trainable parameters a backward inside `body` accumulates into): autograd runs a leaf's AccumulateGrad node on
    # the stream that was current when that node was CREATED, and the node lives as long as any graph that uses the parameter.  A
    # graph kept from an earlier eager step (a held loss / output tensor) therefore drags the stream of that step -- normally the
    # legacy default stream -- into the capture as a forked stream, and hipStreamEndCapture then dies in hip::Stream::EndCapture
    # (seen with a test that held x_hat and the loss across capture_step).  Found out here, during the warm-up: a tensor hook on a
    # leaf runs under that node's stream guard.
    seen, hooks = set(), []
    for p in leaves:
        if p.requires_grad:
            hooks.append(p.register_hook(lambda g, _s=seen: _s.add(torch.cuda.current_stream().cuda_stream) or None))
    try:
        with torch.cuda.stream(side):
            for _ in range(warmup):
                body()
    finally:
        for h in hooks:
            h.remove()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if seen - {side.cuda_stream}:
        raise RuntimeError(f"{what}: an autograd graph from an earlier eager step still references the parameters (their gradient "
                           "accumulation ran on another stream than the warm-up's) -- capturing now would pull that stream into the "
                           "graph and crash in hipStreamEndCapture.  Drop the old loss / output tensors (del them) before capturing.")
    memset_guard(body, what)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        out = body()
    return graph, out


def memset_guard(body, what):
    """A captured hipMemsetAsync was observed NOT to be replayed with the graph on this ROCm stack (libw2e.so zero-fills with
    kernels for that reason).  torch's multi-block reductions zero their scratch with one, so a body that contains a memset
    (e.g. a callable mask with a large .mean()) would replay on stale scratch: run it once under the profiler and refuse it here
    rather than diverge silently.  Fails CLOSED: a profile without any device-side event (the process already runs under
    rocprofv3, or kineto has no roctracer) proves nothing, so the capture is refused then too."""
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
        body()
        torch.cuda.synchronize()
    events = prof.events()
    if not any(e.kernels for e in events):
        raise RuntimeError(f"{what} could not be checked for memset operations: the profiler delivered no device-side events "
                           "(another profiler attached?) -- refusing to capture it; run eagerly (bench.py --graph off)")
    memsets = sorted({e.name for e in events if any("emset" in k.name or "fillBuffer" in k.name for k in (e.kernels or []))})
    if memsets:
        raise RuntimeError(f"{what} issues memset operations (from {', '.join(memsets)}); they are not replayed "
                           "reliably inside a hipGraph here -- use an elementwise / kernel-based form (tools/graph_safety.py lists them)")


def synthetic_tokens(n_text=1, context_length=77, vocab_size=49408, seed=0):
    """Stand-in for clip.tokenize(description): <SOT> ... <EOT> with the EOT id the highest (argmax pooling)."""
    g = torch.Generator().manual_seed(seed)
    t = torch.zeros(n_text, context_length, dtype=torch.int64)
    for i in range(n_text):
        n = 6 + i
        t[i, 0] = vocab_size - 2
        t[i, 1:n] = torch.randint(1, vocab_size - 2, (n - 1,), generator=g)
        t[i, n] = vocab_size - 1
    return t
