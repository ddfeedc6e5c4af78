#!/usr/bin/env python3
"""Headline benchmark: 1024^2 edited images/s per StyleCLIP-mapper training step (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    N>1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or plain `python bench.py --gpus N`: the
    parent then starts N worker processes itself, one per GPU, BEFORE it touches the GPU (the reference's
    mp.spawn, attention/run_attention.py:913-919), relays rank 0's JSON line and exits with the workers' status.

One "step" = one iteration of mapper/training/coach.py:79-92 on this rank's shard of synthetic
FFHQ-shape W+ latents: G(w) [no grad] -> w_hat = w + 0.1 M(w) -> G(w_hat) -> CLIP loss + latent L2 ->
backward -> (all-reduce of mapper grads) -> Ranger step.  Workload at N=1 = BASELINE configs[1]:
FFHQ-1024 StyleGAN2 + clip_loss, batch 4.  N>1 = BASELINE configs[3]: the same step data-parallel at 8 latents per GPU
(64 over 8 GPUs), weak scaling (8 per GPU at every N>1); the N=1 line carries `n1_b8`, the single-GPU figure at that same
per-GPU batch, so that a 1 -> N ratio can be taken at equal per-GPU work.
Weights are random-init of the real architectures (no network for checkpoints), data is synthetic.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the ModulatedConv2d 3x3 convs of the step (direct MFMA kernels + the two Winograd F(4x4,3x3) forms, all own kernels):
                  FLOPs the matrix pipes EXECUTE / HIP-event time vs the fp32-MFMA peak (`frac` <= 1 by construction); `dominant_kernel`
                  = w2e::modconv_kernel alone; `effective_tflops` = the direct-form (algorithmic) FLOPs of the same calls over the same time
                  (the Winograd forms execute 1/4 of them) -- an effective rate, not a roofline fraction
  cpu_baseline -- the CPU oracle (a port of the reference algorithm) timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

G_FWD_GFLOP_1024 = 148.13  # 3x3 modconv stack per image forward (ToRGB's 0.39 is not MFMA work)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md); only quoted with --conv-precision bf16x3
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def make_opts(size, batch, workload=2):
    if workload == 3:  # BASELINE configs[2]: region-attention mask at layer 13 + id_loss
        o = make_opts(size, batch)
        o.id_lambda, o.attention_layer = 0.1, 13
        return o
    return types.SimpleNamespace(
        mapper_type="LevelsMapper", no_coarse_mapper=False, no_medium_mapper=False, no_fine_mapper=False,
        work_in_stylespace=False, stylegan_size=size, checkpoint_path=None, stylegan_weights=None,
        batch_size=batch, test_batch_size=1, learning_rate=0.5, optim_name="ranger", id_lambda=0.0, clip_lambda=1.0,
        latent_l2_lambda=0.8, max_steps=0, description="synthetic prompt")


def build_coach(size, batch, device, data_parallel, clip_backend="hip", workload=2):
    from where2edit_amd.clip_loss import CLIPLoss
    from where2edit_amd.clip_vit import CLIP
    from where2edit_amd.coach import Coach, synthetic_tokens
    from where2edit_amd.styleclip_mapper import StyleCLIPMapper
    torch.manual_seed(0)  # identical replicas on every rank
    opts = make_opts(size, batch, workload)
    net = StyleCLIPMapper(opts)
    if workload == 3:  # same parameters, the generator class that can blend (attention/attention_model.py)
        from where2edit_amd.attention_model import Generator as AttentionGenerator
        dec = AttentionGenerator(size, 512, 8)
        dec.load_state_dict(net.decoder.state_dict(), strict=True)
        net.decoder = dec
    with torch.no_grad():  # the reference inits these to 0; give them values so the fused epilogue paths are live
        for name, p in net.decoder.named_parameters():
            if name.endswith("noise.weight") or name.endswith("activate.bias") or name.endswith("to_rgb1.bias") \
                    or (".to_rgbs." in name and name.endswith(".bias") and p.ndim == 4):
                p.normal_(0, 0.1)
    clip = CLIPLoss(opts, model=CLIP())
    coach = Coach(opts, net=net, clip_loss=clip, text_inputs=synthetic_tokens(1), device=device,
                  data_parallel=data_parallel)
    return coach


@torch.no_grad()
def synthetic_latents(gen, batch, rank):
    """z ~ N(0,I), seed 1234+rank -> style MLP -> truncation 0.7 toward mean_latent(4096) -> W+."""
    dev = gen.input.input.device
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    z = torch.randn(batch, 512, generator=g).to(dev)
    gm = torch.Generator(device="cpu").manual_seed(4096)
    mean_w = gen.style(torch.randn(4096, 512, generator=gm).to(dev)).mean(0, keepdim=True)
    w = mean_w + 0.7 * (gen.style(z) - mean_w)
    return w.unsqueeze(1).repeat(1, gen.n_latent, 1).contiguous()


def make_mask(coach, batch, size, rank, device, synthetic=False):
    """The mask of BASELINE configs[2].  Default: the region-attention net's mask branch (run_attention.py:754-884) on the
    activations of G(w) -- nearest-centroid assignment of the layer-13 features to 20 clusters, the 18 gathered 1x1 StyledConvs +
    the 576->1 one + sigmoid, per-cluster pooling, threshold, 5x5 gaussian; random-init net and centres (no network).  Returned
    as a callable features -> mask (Coach evaluates it between the two generator passes).  `synthetic`: a seeded U(0,1) tensor."""
    if size == 1024 and not synthetic:
        from where2edit_amd.run_attention import FullSpaceMapperFEATClusterLinStyle_Net, cluster_pool
        torch.manual_seed(1)
        att_net = FullSpaceMapperFEATClusterLinStyle_Net(18, 1024, 512, attention_layer=13, cluster_layer=13, channel_multiplier=2,
                                                         clusters=20, cluster_dim=576).to(device).requires_grad_(False)
        with torch.no_grad():  # (the reference's init of 5 saturates the sigmoid: every cluster passes the threshold; logit(0.8)
            att_net.initial_bias.fill_(1.3863)  # puts the random-init net's cluster means on both sides of it: a mixed mask)
        att_text = torch.randn(batch, 512, generator=torch.Generator().manual_seed(5 + rank)).to(device) * 0.3
        const_in = coach.net.decoder.input.input

        def mask(feats):
            fm = list(feats) + [const_in.repeat(batch, 1, 1, 1)]
            if not hasattr(att_net, "_seeded"):  # centres = 20 pixels of the first batch's features: non-trivial clusters
                f = fm[12]
                idx = torch.randperm(64 * 64, generator=torch.Generator().manual_seed(3))[:20]
                pts = f[0].reshape(512, -1)[:, idx.to(f.device)].t()
                ys, xs = (idx // 64).float() * 2 / 63 - 1, (idx % 64).float() * 2 / 63 - 1
                att_net.store_clusters(torch.cat([pts, xs.to(f.device)[:, None].repeat(1, 32), ys.to(f.device)[:, None].repeat(1, 32)], 1))
            att_net._seeded = True
            each, assign = att_net.attention_map(fm, 64, att_text, 26)
            mask.last = cluster_pool(each, assign, 64, 20)[4]
            return mask.last
        return mask
    res = 4 * 2 ** ((13 - 1) // 3) if size == 1024 else max(4, size // 16)  # seeded U(0,1) at the resolution of layer 13
    return torch.rand(batch, 1, res, res, generator=torch.Generator().manual_seed(77 + rank)).to(device)


def pmc_traffic():
    """HBM bytes per launch of the conv kernel from the rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of
    this same command, corrected as MI355X_MICROARCH.md prescribes) -- recorded in profiles/traffic.json by
    tools/pmc_traffic.py; None when that file is absent."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["modconv_hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(size):
    """The oracle's mapper step (oracle/step.py, a port of coach.py:79-92 on stock CPU torch ops) at batch 1
    on this box's host cores: one full step (2 G forwards + CLIP + backward + Ranger).  Bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import seeded
    from oracle import step as ostep
    from oracle import stylegan2 as og
    # threads = this process's CPU share: the affinity mask, capped at the 16 host cores a one-GPU box is given
    # (os.cpu_count() reports the whole 256-thread host there and oversubscribing it is ~50x slower)
    cores = int(os.environ.get("W2E_CPU_THREADS", min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16)))
    torch.set_num_threads(cores)
    gsd = seeded.generator_state_dict(size)
    msd = {k: v.requires_grad_(True) for k, v in
           seeded.mapper_state_dict(["course_mapping.", "medium_mapping.", "fine_mapping."]).items()}
    csd = seeded.clip_state_dict()
    from where2edit_amd.coach import synthetic_tokens
    tokens = synthetic_tokens(1)
    w = seeded.wplus_latents(1, og.n_latent(size), salt=7)
    params = list(msd.values())
    st = ostep.RangerState(params, lr=0.5)
    with torch.no_grad():  # page in the weights / spin up the thread pool on a small case, untimed
        og.generator_forward(seeded.generator_state_dict(64), [seeded.wplus_latents(1, 10)], size=64, input_is_latent=True,
                             randomize_noise=False)
    n_steps = 3
    t0 = time.perf_counter()
    for _ in range(n_steps):
        loss, _, _, _, _ = ostep.mapper_step_loss(gsd, msd, csd, w, tokens, size=size, clip_lambda=1.0, latent_l2_lambda=0.8)
        grads = torch.autograd.grad(loss, params)
        st.step(params, grads)
    dt = time.perf_counter() - t0
    # one more step on ONE thread (SURVEY 8d asks for the single-core figure beside the all-core one), and
    # BASELINE configs[0] (the reference's own CPU-runnable case: 256^2 generator forward, one latent) on all cores
    torch.set_num_threads(1)
    t1 = time.perf_counter()
    loss, _, _, _, _ = ostep.mapper_step_loss(gsd, msd, csd, w, tokens, size=size, clip_lambda=1.0, latent_l2_lambda=0.8)
    st.step(params, torch.autograd.grad(loss, params))
    dt1 = time.perf_counter() - t1
    torch.set_num_threads(cores)
    g256, w256 = seeded.generator_state_dict(256), seeded.wplus_latents(1, og.n_latent(256), salt=3)
    with torch.no_grad():
        og.generator_forward(g256, [w256], size=256, input_is_latent=True, randomize_noise=False)
        t2 = time.perf_counter()
        og.generator_forward(g256, [w256], size=256, input_is_latent=True, randomize_noise=False)
        dt2 = time.perf_counter() - t2
    return {"value": n_steps / dt, "unit": "images/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "one_thread": {"value": 1.0 / dt1, "unit": "images/s", "cores": 1, "sample": f"1 mapper step, batch 1, {size}^2, {dt1:.1f} s"},
            "config0_g256_forward": {"value": 1.0 / dt2, "unit": "images/s", "cores": cores,
                                     "sample": f"StyleGAN2-256 generator forward, 1 latent (BASELINE configs[0]), {dt2:.2f} s"},
            "sample": f"{n_steps} mapper steps (each 2 G fwd + CLIP ViT-B/32 fwd/bwd + G bwd + Ranger), batch 1, {size}^2, "
                      f"oracle/ on torch {torch.__version__} CPU ops, {cores} threads, {dt:.1f} s"}


def self_launch(args):
    """`python bench.py --gpus N` with no torchrun environment: start the N ranks from here.  Nothing in this process has
    touched the GPU (no HIP call, not even torch.cuda.is_available()), and the workers are fresh interpreters -- a process
    that has initialised the GPU is never replaced or forked."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver supports dmabuf IPC only; without it RCCL's intra-node
        # transport fails in hipIpcGetMemHandle ("invalid argument").  The image exports it already -- kept (not overridden)
        # here so that the ranks get it even from a stripped environment.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread while ALL ranks are polled: a rank that dies at start-up would otherwise leave
    # the others in init_process_group / the first all-reduce until the backend's timeout, and this parent with them
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with status {failed[1]}; the other ranks were stopped\n")
    reader.join(timeout=10.0)
    rcs = [p.wait() for p in procs]
    out = chunks[0] if chunks else b""
    for line in out.decode().splitlines():  # ONE JSON line on stdout (the contract); anything else rank 0 printed -> stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    raise SystemExit(max(abs(rc) for rc in rcs))


def rehearse(args):
    """`--rehearse`: the control flow of an N-rank run of main() with NO GPU work, so that the launcher, the rendezvous, the collective
    stop rule of stabilise(), the barriers, the MAX over ranks and the JSON relay have run at the world size a real node will use
    before a real node sees them (a GPU box allows at most 6 processes on its card; this needs none).  The step is a stand-in: a
    2 ms sleep whose length depends on the rank (so that the ranks would leave the stabilisation loop after different numbers of
    steps if they decided alone) + the all-reduce of a LevelsMapper-sized bucket (3.15 M floats) of host memory over gloo."""
    from where2edit_amd import dist as wd
    rank, world, _ = wd.init_from_env(backend="gloo", timeout_s=120, use_gpu=False)  # (no rank may open the GPU: see its docstring)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    bucket = torch.zeros(3_150_000)
    calls = [0]

    def step():
        calls[0] += 1
        time.sleep(0.002 * (1.0 + 0.5 * rank / max(world - 1, 1)) * (3.0 if calls[0] <= 2 + rank else 1.0))  # rank-dependent settling
        bucket.fill_(float(rank + 1))
        if world > 1:
            torch.distributed.all_reduce(bucket)
            bucket.div_(world)
        return {"loss": bucket[0]}

    sync_word = torch.zeros(1)

    def barrier():  # (an all-reduce of a host word: torch.distributed.barrier() picks a device and OPENS the GPU even on gloo -- seen on a GPU
        if world > 1:  # box, where 8 rehearsal ranks then ran into the 6-processes-per-card guard)
            torch.distributed.all_reduce(sync_word)

    n_stab, ok = stabilise(step, world=world, device="cpu", sync=lambda: None, max_seconds=10.0)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    want = sum(range(1, world + 1)) / world  # every rank's bucket holds the mean of (rank + 1)
    if abs(float(last["loss"]) - want) > 1e-6:
        raise SystemExit(f"rehearsal: the bucket holds {float(last['loss'])} after the all-reduce, expected {want}")
    if world > 1:
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "rehearsal of the N-rank launcher (no GPU work, stand-in step): NOT a measurement", "value": None,
                          "unit": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                          "rehearsal": True, "steps_run_per_rank": calls[0], "stabilise_steps": n_stab, "stabilised": ok,
                          "config": {"workload": "stand-in step: sleep + gloo all-reduce of a 12.6 MB host bucket", "dist_backend": "gloo"}}),
              flush=True)


def stabilise(step_fn, tol=0.02, window=5, max_steps=80, max_seconds=20.0, world=1, device=None, sync=None):
    """Untimed full steps until the last `window` step times agree within `tol` ((max-min)/mean): a fresh box leaves its idle
    clocks, caches / allocator pools / lazily built packs settle.  Independent of --warmup, so the timed region starts at
    steady state whatever the caller asks for (round 1 lost 10 % of its headline to a 5-step warm-up).
    With several ranks the decision to stop is taken TOGETHER (every step holds a collective: ranks that left the loop after
    different numbers of steps would leave the last all-reduces of the slower ones without a partner): stop when every rank is
    stable, or when any rank has hit a limit."""
    times = []
    sync = sync if sync is not None else torch.cuda.synchronize
    t_begin = time.perf_counter()
    while True:
        sync()
        t0 = time.perf_counter()
        step_fn()
        sync()
        times.append(time.perf_counter() - t0)
        last = times[-window:]
        stable = len(last) == window and (max(last) - min(last)) <= tol * (sum(last) / window)
        limit = len(times) >= max_steps or time.perf_counter() - t_begin >= max_seconds
        if world > 1:
            flags = torch.tensor([1.0 if stable else 0.0, 0.0 if limit else 1.0], device=device)
            torch.distributed.all_reduce(flags, op=torch.distributed.ReduceOp.MIN)
            stable, limit = bool(flags[0].item() == 1.0), bool(flags[1].item() == 0.0)
        if stable or limit:
            return len(times), stable


def build_config5(device, batch, rank):
    """The modules and inputs of BASELINE configs[4] (random-init weights of the real architectures, seeded inputs): returns
    (imgs [B,3,256,256], e4e, generator, clip_loss, region-attention net, text features, attention-text features)."""
    from where2edit_amd.attention_model import Generator
    from where2edit_amd.clip_loss import CLIPLoss
    from where2edit_amd.clip_vit import CLIP
    from where2edit_amd.psp_encoders import Encoder4Editing
    from where2edit_amd.run_attention import FullSpaceMapperFEATClusterLinStyle_Net
    torch.manual_seed(0)
    opts = types.SimpleNamespace(stylegan_size=1024)
    g = Generator(1024, 512, 8).to(device).eval().requires_grad_(False)
    e4e = Encoder4Editing(50, "ir_se", opts).to(device).eval().requires_grad_(False)
    clip = CLIPLoss(opts, model=CLIP()).to(device)
    net = FullSpaceMapperFEATClusterLinStyle_Net(18, 1024, 512, attention_layer=13, cluster_layer=13, channel_multiplier=2,
                                                 clusters=20, cluster_dim=576).to(device).eval().requires_grad_(False)
    gen = torch.Generator().manual_seed(100 + rank)
    imgs = (torch.rand(batch, 3, 256, 256, generator=gen) * 2 - 1).to(device)
    text, att = (torch.randn(batch, 512, generator=gen) * 0.3).to(device), (torch.randn(batch, 512, generator=gen) * 0.3).to(device)
    with torch.no_grad():  # random-init net: logit(0.8) as the bias and 20 pixels of the first batch's layer-13 features as the
        net.initial_bias.fill_(1.3863)  # cluster centres give a mixed (not all-0 / all-1) thresholded mask, as in make_mask()
        _, _, styles0 = g([e4e(imgs)], input_is_latent=True, return_latents=True, randomize_noise=False)
        _, _, _, feats0 = g([styles0], input_is_latent=True, randomize_noise=False, return_features=True, input_is_stylespace=True)
        f = feats0[12]
        idx = torch.randperm(64 * 64, generator=torch.Generator().manual_seed(3))[:20]
        pts = f[0].reshape(512, -1)[:, idx.to(f.device)].t()
        ys, xs = (idx // 64).float() * 2 / 63 - 1, (idx % 64).float() * 2 / 63 - 1
        net.store_clusters(torch.cat([pts, xs.to(f.device)[:, None].repeat(1, 32), ys.to(f.device)[:, None].repeat(1, 32)], 1))
        del styles0, feats0, f
    return imgs, e4e, g, clip, net, text, att


def measure_config5(args, rank, world, device, batch):
    """BASELINE configs[4] on this rank's share of the images: 256^2 image -> e4e -> S codes -> features -> region-attention
    net (mask + new codes) -> masked 1024^2 generator -> CLIP features (where2edit_amd.demo_pipeline.invert_and_edit).
    Inference: every rank works on its own images, nothing is exchanged.  Random-init weights of the real architectures."""
    from where2edit_amd.demo_pipeline import invert_and_edit
    imgs, e4e, g, clip, net, text, att = build_config5(device, batch, rank)

    def step():
        return invert_and_edit(imgs, e4e, g, clip, net, text, att, attention_layer=13)

    graph_note, use_graph = None, False
    if args.graph != "off":  # the fixed-shape pipeline as one hipGraph (about 1500 launches per call: host-bound when enqueued eagerly)
        try:
            from where2edit_amd.demo_pipeline import capture_invert_and_edit
            graphed = capture_invert_and_edit(imgs, e4e, g, clip, net, text, att, attention_layer=13)
            step = lambda: graphed(imgs, text, att)  # noqa: E731
            use_graph = True
        except Exception as e:  # noqa: BLE001
            if args.graph == "on":
                raise
            graph_note = f"eager: capture failed ({type(e).__name__}: {e})"[:300]
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    n_stab, ok = stabilise(step)  # (replicas: no collective inside a step, every rank may stop by itself)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    finite = bool(torch.isfinite(out["img_gen"]).all())
    if not finite:
        raise SystemExit("non-finite output")
    return {"value": batch * world * args.steps / dt, "unit": "images/s", "ms_per_step": 1e3 * dt / args.steps, "batch": batch,
            "workload": f"BASELINE configs[4]: e4e encode -> S codes -> 26 features -> region-attention net (cluster-pooled mask, new codes) -> "
                        f"masked FFHQ-1024 generator -> CLIP features (show_demo/try_demo.py:93-157), batch {batch}/GPU, no backward",
            "stabilise_steps": n_stab, "stabilised": ok, "mask_mean": float(out["mask"].mean()), "hip_graph": use_graph,
            "hip_graph_note": graph_note}


def bench_config5(args, rank, world, device):
    r = measure_config5(args, rank, world, device, args.batch)
    if world > 1:
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    print(json.dumps({
        "metric": "1024^2 edited images/sec, invert-and-edit inference pipeline (BASELINE configs[4])", "value": r["value"],
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": r["workload"], "global_batch": args.batch * world, "per_gpu_batch": args.batch, "parallelism": f"replicas x{world}",
                   "stabilise_steps": r["stabilise_steps"], "stabilised": r["stabilised"], "mask_mean": r["mask_mean"],
                   "hip_graph": r["hip_graph"], "hip_graph_note": r["hip_graph_note"]}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~0.7 s of warm-up (a fresh box needs that long to bring the GPU out of its idle clocks: 3 warm-up steps
    # measured 131 images/s as the first process on a box, 25 measured 142) and ~0.6 s timed
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--batch", type=int, default=None,
                    help="latents per GPU (weak scaling).  Default: 4 at --gpus 1 (BASELINE configs[1]), 8 at --gpus N > 1 "
                         "(configs[3]: 64 latents over 8 GPUs)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-stride", type=int, default=4,
                    help="the roofline's HIP events are recorded on every N-th timed step (166 timed event records per step "
                         "cost the step 5 %%: 159 images/s without them, 150 with them on every step)")
    ap.add_argument("--no-preview", action="store_true", help="skip the extra K steps in the opt-in bf16x3 conv precision (N=1 only)")
    ap.add_argument("--side-stream", action="store_true",
                    help="inside the graph, fork the no-grad G(w) pass onto a second stream (+1-2 %% images/s; off by default: "
                         "overlapped kernels stretch each other's durations, so a rocprofv3 trace of the run would no longer "
                         "show the per-kernel times the roofline is quoted on)")
    ap.add_argument("--no-n1-b8", action="store_true", help="skip the extra batch-8 measurement (`n1_b8`) of a default N=1 run")
    ap.add_argument("--no-config3", action="store_true", help="skip the extra BASELINE configs[2] measurement of a default N=1 run")
    ap.add_argument("--no-config5", action="store_true", help="skip the extra BASELINE configs[4] measurement (batch 4 = 32 images / 8 GPUs) of a default N=1 run")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"], nargs="?", const="on",
                    help="replay the step as one captured hipGraph (Coach.capture_step) instead of ~330 eager launches; the per-kernel "
                         "HIP-event roofline is then taken from eager steps run right after the timed region (same kernels, same "
                         "stream order).  auto (default): graph when the capture succeeds and the mask is a tensor, else eager; "
                         "off: eager")
    ap.add_argument("--synthetic-mask", action="store_true", help="workload 3: a seeded U(0,1) mask instead of the region-attention net's")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank; default) or gloo (self-test: the ranks may share a GPU)")
    ap.add_argument("--no-stabilise", action="store_true", help="skip the untimed steady-state loop before the warm-up steps")
    ap.add_argument("--conv-precision", default="f32", choices=["f32", "bf16x3"],
                    help="f32 (default): exact fp32 MFMA everywhere.  bf16x3: the SAME-resolution and up-sampling conv tiles compute each fp32 "
                         "product as three bf16 products (hi*hi + hi*lo + lo*hi, ~2^-17 relative error per product; every "
                         "parity test passes with it) -- opt-in, reported as dtype bf16x3")
    ap.add_argument("--clip-precision", default="f32", choices=["f32", "f16"],
                    help="f32 (default): the CLIP tower on exact fp32 MFMA.  f16: the four Linear layers of every block take fp16 operands "
                         "(fp32 accumulation) -- the arithmetic of the fp16 model the reference loads on a GPU (criteria/clip_loss.py:10); "
                         "opt-in, recorded in config.clip_precision and in dtype; the default line only carries it as `clip_f16_preview`")
    ap.add_argument("--workload", type=int, default=2, choices=[2, 3, 5],
                    help="BASELINE configs index + 1: 2 = clip_loss mapper step (the headline, default); 3 = the same step with "
                         "the region-attention mask blend at layer 13 and id_loss (quoted at batch 8); 5 = the inference pipeline "
                         "e4e encode -> cluster-pooled mask -> mapper edit -> 1024^2 generator (show_demo/try_demo.py:93-157; "
                         "replicas only at N > 1, no collective)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --batch latents per GPU at every N (8 at N > 1: BASELINE configs[3] at N = 8).  strong: the "
                         "GLOBAL batch is fixed at --global-batch (64 = configs[3]) whatever N; a GPU's shard of 64 / N latents runs as "
                         "micro-batches of --batch (8) whose gradients are accumulated before the one all-reduce and the one optimizer "
                         "step (Coach.accumulated_step) -- SURVEY 8(d) cfg4 asks for both curves")
    ap.add_argument("--global-batch", type=int, default=64, help="--scaling strong: latents per step over all GPUs")
    ap.add_argument("--no-n1-reference", action="store_true",
                    help="N > 1: skip the K extra steps WITHOUT the gradient all-reduce that give the line its one-GPU figure at "
                         "the same per-GPU batch (`n1_equal_batch`)")
    ap.add_argument("--lib-option", action="append", default=[], metavar="NAME=VALUE",
                    help="w2e_set_option(NAME, VALUE) before the run -- for same-box A/Bs of a kernel switch inside the real step (e.g. "
                         "tune_mw=4: the fused Winograd kernel's 32-channel workgroups; tune_xcd=0: its old block order); recorded in "
                         "config.lib_options, so a line measured with one says so.  (The W2E_TUNE_* environment variables stay refused.)")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher rehearsal WITHOUT any GPU work (runs where there is no GPU, at any world size): the real control flow "
                         "of an N-rank run -- self-launch or torchrun environment, gloo rendezvous, the collective stop rule of the "
                         "stabilisation loop, warm-up, barriers, K timed steps, MAX over ranks, rank 0's one JSON line -- around a "
                         "stand-in step (2 ms sleep + the all-reduce of a 12.6 MB bucket of host memory).  The line it prints says so "
                         "and carries no value")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 4 if (args.gpus == 1 and args.scaling == "weak") else 8
    if args.scaling == "strong":
        if args.workload != 2:
            raise SystemExit("--scaling strong is defined for the mapper training step (--workload 2)")
        if args.global_batch % (args.gpus * args.batch):
            raise SystemExit(f"--scaling strong: global batch {args.global_batch} is not a whole number of micro-batches of "
                             f"{args.batch} on each of {args.gpus} GPUs")
    tune = sorted(k for k in os.environ if k.startswith("W2E_TUNE_"))
    if tune:  # the tuning aids can skip work or force slow tiles: never measure with them set
        raise SystemExit(f"bench.py refuses to run with tuning variables set: {', '.join(tune)}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    if args.rehearse:
        return rehearse(args)
    from where2edit_amd import _lib
    from where2edit_amd import dist as wd
    from where2edit_amd import profiling
    rank, world, local = wd.init_from_env(backend=args.dist_backend, timeout_s=300)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and world > n_dev:
        raise SystemExit(f"--gpus {world} but this node has {n_dev} GPU(s) (RCCL wants one device per rank; --dist-backend gloo "
                         "lets ranks share a GPU for a self-test)")
    device = f"cuda:{local % max(n_dev, 1)}"
    torch.cuda.set_device(device)
    _lib.set_option("conv_precision", args.conv_precision)
    for kv in args.lib_option:
        name, _, val = kv.partition("=")
        _lib.set_option(name, val)
    if args.workload == 5:
        return bench_config5(args, rank, world, device)
    strong = args.scaling == "strong"
    per_gpu = args.global_batch // world if strong else args.batch  # latents this GPU processes per step
    n_micro = per_gpu // args.batch                                  # ... as this many passes of --batch latents (1 unless strong)
    coach = build_coach(args.size, args.batch, device, world > 1 or strong, 'hip', args.workload)
    if args.clip_precision != "f32":
        coach.clip_loss.model.set_precision(args.clip_precision)
    w_all = synthetic_latents(coach.net.decoder, per_gpu, rank)
    chunks = list(w_all.split(args.batch))
    w = chunks[0]
    mask = make_mask(coach, args.batch, args.size, rank, device, args.synthetic_mask) if args.workload == 3 else None

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    eager_fn = (lambda: coach.accumulated_step(chunks)) if strong else (lambda: coach.train_step(w, mask))  # noqa: E731
    step_fn = eager_fn
    use_graph, graph_note, graphed = False, None, None
    if args.graph != "off":
        try:  # (a callable mask -- the region-attention net's mask branch -- is captured between the two generator passes)
            graphed = coach.capture_step(w, mask, side_stream=args.side_stream)
            step_fn = (lambda: coach.accumulated_step(chunks, graphed)) if strong else (lambda: graphed(w, mask))  # noqa: E731
            use_graph = True
        except Exception as e:  # noqa: BLE001  (any capture failure: measure eagerly rather than not at all)
            if args.graph == "on":
                raise
            graph_note = f"eager: capture failed ({type(e).__name__}: {e})"[:300]
            torch.cuda.synchronize()
    args.graph = use_graph
    stab_steps, stab_ok = (0, False) if args.no_stabilise else stabilise(step_fn, world=world, device=device)
    for _ in range(args.warmup):
        step_fn()
    barrier()
    timer = None if args.no_kernel_timing else profiling.KernelTimer()
    sampled = 0
    if args.graph:  # the timed region is graph replays only; the per-kernel roofline comes from eager steps after it
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = step_fn()
        barrier()
        dt = time.perf_counter() - t0
        if timer is not None:
            timer.__enter__()
            for i in range(max(1, args.steps // args.timing_stride)):
                eager_fn()
                sampled += 1
            barrier()
            timer.__exit__(None, None, None)
    else:
        if timer is not None:
            timer.__enter__()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if timer is not None:  # HIP events around the conv / blur launches on every `--timing-stride`-th timed step
                timer.enabled = i % args.timing_stride == 0
                sampled += int(timer.enabled)
            last = step_fn()
        barrier()
        dt = time.perf_counter() - t0
        if timer is not None:
            timer.__exit__(None, None, None)
    n1_ref = None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
        if not args.no_n1_reference:
            # The one-GPU figure this line's `value` has to be divided by for a scaling ratio: the SAME K steps at the SAME per-GPU
            # batch, same process, same box, with the gradient all-reduce taken out -- i.e. what one GPU does when it is alone
            # (GradBucket.all_reduce_mean is the step's only collective).  Every rank runs them; the slowest rank's time is used.
            bucket_reduce, coach.bucket.all_reduce_mean = coach.bucket.all_reduce_mean, (lambda: None)
            try:
                for _ in range(3):
                    step_fn()
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step_fn()
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
            finally:
                coach.bucket.all_reduce_mean = bucket_reduce
            t = torch.tensor([dt1, -dt1], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            n1_ref = {"value": per_gpu * args.steps / t[0].item(), "unit": "images/s", "ms_per_step": 1e3 * t[0].item() / args.steps,
                      "fastest_rank_ms_per_step": 1e3 * -t[1].item() / args.steps, "per_gpu_batch": per_gpu,
                      "what": f"ONE GPU's rate on this step at {per_gpu} latents per GPU: the same {args.steps} steps in the same processes "
                              "right after the timed region with the gradient all-reduce removed (slowest rank); "
                              "value / (n_gpus x this) = the parallel efficiency at equal per-GPU work"}
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    loss = float(last["loss"])
    if not (loss == loss):
        raise SystemExit("loss is NaN")
    global_batch = per_gpu * world
    value = global_batch * args.steps / dt
    out = {
        "metric": "1024^2 edited images/sec per mapper step" if args.size == 1024 else f"{args.size}^2 edited images/sec per mapper step",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.conv_precision + ("" if args.clip_precision == "f32" else f" (CLIP block GEMM operands {args.clip_precision})"), "data": "synthetic",
        "config": {"workload": ((f"BASELINE configs[1]: " if (world == 1 and per_gpu == 4 and args.size == 1024) else
                                 f"BASELINE configs[3] (FFHQ-1024 mapper training, global batch {global_batch}, data-parallel over {world} GPUs, "
                                 f"RCCL all-reduce of the mapper gradients): " if (global_batch == 64 and args.size == 1024) else
                                 f"the per-GPU workload of BASELINE configs[3] on {world} GPU(s) (weak scaling towards 64 latents on 8): "
                                 if (per_gpu == 8 and args.size == 1024) else "") +
                                f"FFHQ-{args.size} StyleGAN2 + clip_loss mapper step (coach.py:79-92), {per_gpu} latents/GPU" +
                                (f" as {n_micro} micro-batches of {args.batch} with accumulated gradients" if n_micro > 1 else "") +
                                f", LevelsMapper, Ranger, id_lambda=0" +
                                (f"; scaling reference = `n1_equal_batch` of this line (one GPU at {per_gpu} latents, no collective), "
                                 f"same workload as `bench.py --gpus 1 --batch {per_gpu}` / `n1_b8` of the N=1 line" if (world > 1 and not strong) else "") +
                                (f"; strong scaling: global batch {global_batch} at every N, the N=1 point is `bench.py --gpus 1 --scaling strong`"
                                 if strong else "")) if args.workload == 2 else
                               (f"FFHQ-{args.size} mapper step with the region-attention mask (cluster-pooled, thresholded, blurred: run_attention.py:754-884"
                                f"{' -- synthetic U(0,1) mask' if callable(mask) is False else ''}) blended at layer 13 (attention_model.py) "
                                f"+ clip_loss + id_loss (IR-SE50), batch {args.batch}/GPU, LevelsMapper, Ranger, id_lambda=0.1"),
                   "global_batch": global_batch, "per_gpu_batch": per_gpu, "micro_batch": args.batch, "micro_batches_per_step": n_micro,
                   "parallelism": f"dp{world}", "conv_precision": args.conv_precision, "final_loss": loss,
                   "stabilise_steps": stab_steps, "stabilised": stab_ok, "hip_graph": bool(args.graph), "hip_graph_note": graph_note, "mask_mean": (float(mask.last.mean()) if hasattr(mask, "last") else None), "side_stream": bool(args.side_stream and args.graph), "dist_backend": args.dist_backend if world > 1 else None,
                   "lib_options": args.lib_option or None, "clip_precision": args.clip_precision},
    }
    if n1_ref is not None:
        out["n1_equal_batch"] = n1_ref
        out["parallel_efficiency_vs_n1_equal_batch"] = value / (world * n1_ref["value"])
    if timer is not None:
        s = timer.summary()
        # every 3x3 modulated conv of the step: the direct MFMA kernels ("modconv3x3") and the Winograd-form layers
        # ("modconv3x3_wino<m>": input transform + library GEMM + output transform inside one span).  `achieved` is what the contract
        # asks for -- ALGORITHMIC (direct-form) FLOPs over the measured time; the Winograd forms execute (m+2)^2 / (9 m^2)
        # of their algorithmic FLOPs (1/4 for F(4x4,3x3)), so `achieved` is an effective rate there: `executed` is what the matrix
        # pipes really did, `direct` the MFMA kernels alone.
        d_calls, d_ms, d_flops = s.get("modconv3x3", (0, 0.0, 0.0))
        w_calls, w_ms, w_flops, w_exec = 0, 0.0, 0.0, 0.0
        for m_ in (2, 4):
            c_, ms_, fl_ = s.get("modconv3x3_wino%d" % m_, (0, 0.0, 0.0))
            w_calls, w_ms, w_flops, w_exec = w_calls + c_, w_ms + ms_, w_flops + fl_, w_exec + fl_ * (m_ + 2) ** 2 / (9.0 * m_ * m_)
        calls, ms, flops = d_calls + w_calls, d_ms + w_ms, d_flops + w_flops
        if calls:
            # Contract fields: `achieved` / `frac` price the conv family on the FLOPs its kernels EXECUTE (a Winograd
            # form executes (m+2)^2/(9 m^2) of the direct form's), so frac <= 1 by construction; the direct-form ("algorithmic",
            # SURVEY 8d) FLOPs over the same time are an effective rate and live under `effective_tflops`, never under `frac`.
            # `dominant_kernel` = the single kernel with the largest share of the step (w2e::modconv_kernel: its algorithmic
            # and executed FLOPs are the same thing).
            executed = (d_flops + w_exec) / (ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "every 3x3 modulated conv call of the step, fwd + dgrad: w2e::modconv_kernel (fp32 MFMA "
                               "32x32x2 implicit GEMM: up-sampling convs, their stride-2 adjoints, <= 8^2 layers)" +
                               (", the fused F(4x4,3x3) kernel (w2e::wino4_fused3_kernel) and the Winograd-domain contraction kernel "
                                "(w2e::wino4_gemm_kernel: w2e::wino4_pack_input_kernel in front of it, the output transform in its epilogue)" if w_calls else ""),
                               "achieved": executed, "peak": FP32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": executed / FP32_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(),
                               "flops": "EXECUTED by the matrix pipes: direct form 2*K*N*9 per pixel; an F(4x4,3x3) call 36/144 of that",
                               "launches": calls, "avg_launch_ms": ms / calls, "flop_per_launch": (d_flops + w_exec) / calls,
                               "share_of_step": (ms / max(sampled, 1)) / (1e3 * dt / args.steps),
                               "timed_steps": sampled,
                               "effective_tflops": flops / (ms * 1e-3) / 1e12,
                               "effective_note": "algorithmic (direct-form, SURVEY 8d) FLOPs of the same calls over the same time; exceeds the "
                                                 "executed rate because the Winograd forms do fewer multiplications -- not a roofline fraction",
                               "dominant_kernel": ({"name": "w2e::modconv_kernel", "launches": d_calls, "avg_launch_ms": d_ms / d_calls,
                                                    "flop_per_launch": d_flops / d_calls, "achieved": d_flops / (d_ms * 1e-3) / 1e12,
                                                    "frac": d_flops / (d_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                                    "share_of_step": (d_ms / max(sampled, 1)) / (1e3 * dt / args.steps)} if d_calls else None),
                               "winograd": ({"calls": w_calls, "avg_call_ms": w_ms / w_calls, "achieved": w_exec / (w_ms * 1e-3) / 1e12,
                                             "frac": w_exec / (w_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                             "effective_tflops": w_flops / (w_ms * 1e-3) / 1e12,
                                             "form": str(sys.modules["where2edit_amd.functional"].WINOGRAD)} if w_calls else None)}
            achieved = flops / (ms * 1e-3) / 1e12
            if args.conv_precision == "bf16x3":  # opt-in: algorithmic (fp32-conv) FLOPs against the bf16 matrix peak
                out["roofline"].update({
                    "kernel": "w2e::modconv_kernel (SAME, all-phase UP and DOWN tiles: fp32 as three bf16 products on v_mfma_f32_32x32x16_bf16, "
                              "3.33 issued bf16 FLOPs per algorithmic FLOP; low-resolution tiles: fp32 MFMA 32x32x2)",
                    "peak": BF16_MFMA_PEAK_TFLOPS, "frac": achieved / BF16_MFMA_PEAK_TFLOPS, "traffic": None})
        c2, ms2, by2 = s.get("upfirdn2d", (0, 0.0, 0.0))
        if c2:
            out["roofline_hbm"] = {"bound": "hbm", "kernel": "w2e::upfirdn_*", "achieved": by2 / (ms2 * 1e-3) / 1e9,
                                   "peak": 8000.0, "unit": "GB/s", "frac": by2 / (ms2 * 1e-3) / 1e9 / 8000.0,
                                   "launches": c2, "share_of_step": (ms2 / max(sampled, 1)) / (1e3 * dt / args.steps)}
        # whole-stack figure the north_star target is quoted on: 3 G-equivalents per image per step
        if args.size == 1024:
            stack_tflops = 3 * G_FWD_GFLOP_1024 * 1e9 * global_batch * args.steps / dt / 1e12 / world
            out["stack_mfma_frac_of_step"] = stack_tflops / FP32_MFMA_PEAK_TFLOPS
    if world == 1 and not strong and args.conv_precision == "f32" and not args.no_preview:
        # the same K steps once more with the opt-in conv precision (w2e_set_option):
        # reported beside the headline, never as `value`
        _lib.set_option("conv_precision", "bf16x3")
        for _ in range(5):
            coach.train_step(w, mask)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            last2 = coach.train_step(w, mask)
        barrier()
        dt2 = time.perf_counter() - t1
        _lib.set_option("conv_precision", "f32")
        out["bf16x3_preview"] = {"value": global_batch * args.steps / dt2, "unit": "images/s", "ms_per_step": 1e3 * dt2 / args.steps,
                                 "dtype": "bf16x3", "final_loss": float(last2["loss"]),
                                 "note": "opt-in --conv-precision bf16x3: each fp32 product of the 3x3 convs as three bf16 MFMA products "
                                         "(all parity tests pass with it; DESIGN.md section 7); not the headline"}
    if world == 1 and not strong and args.conv_precision == "f32" and args.clip_precision == "f32" and not args.no_preview and coach is not None:
        # and once more with the opt-in fp16 operands in the CLIP tower's block GEMMs (what the reference's GPU tower computes): beside the
        # headline, never `value`
        coach.clip_loss.model.set_precision("f16")
        for _ in range(5):
            coach.train_step(w, mask)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            last3_ = coach.train_step(w, mask)
        barrier()
        dt3_ = time.perf_counter() - t1
        coach.clip_loss.model.set_precision("f32")
        out["clip_f16_preview"] = {"value": global_batch * args.steps / dt3_, "unit": "images/s", "ms_per_step": 1e3 * dt3_ / args.steps,
                                   "final_loss": float(last3_["loss"]), "eager": True,
                                   "note": "opt-in --clip-precision f16 (eager steps, like bf16x3_preview): the CLIP tower's four Linear layers per block "
                                           "on fp16 operands with fp32 accumulation, criteria/clip_loss.py:10's GPU arithmetic; not the headline"}
    if world == 1 and not strong and args.workload == 2 and args.size == 1024 and args.batch == 4 and args.conv_precision == "f32" and not args.no_n1_b8:
        # the same step at 8 latents per GPU = the per-GPU workload of `--gpus N > 1` (BASELINE configs[3]), so that the driver's
        # N-GPU values have a 1-GPU figure at equal per-GPU batch beside the configs[1] headline.  Never `value`.
        del coach
        torch.cuda.empty_cache()
        coach8 = build_coach(args.size, 8, device, False, "hip", 2)
        w8 = synthetic_latents(coach8.net.decoder, 8, rank)
        step8, graph8 = (lambda: coach8.train_step(w8)), False
        if use_graph:
            try:
                g8 = coach8.capture_step(w8)
                step8, graph8 = (lambda: g8(w8)), True
            except Exception:  # noqa: BLE001
                torch.cuda.synchronize()
        n8, ok8 = stabilise(step8)
        barrier()
        t8 = time.perf_counter()
        for _ in range(args.steps):
            last8 = step8()
        barrier()
        dt8 = time.perf_counter() - t8
        out["n1_b8"] = {"value": 8 * args.steps / dt8, "unit": "images/s", "ms_per_step": 1e3 * dt8 / args.steps, "batch": 8,
                        "final_loss": float(last8["loss"]), "stabilise_steps": n8, "hip_graph": graph8,
                        "workload": "the headline step at 8 latents/GPU on 1 GPU (the per-GPU workload of BASELINE configs[3] / of "
                                    "`bench.py --gpus N` for N > 1); same as `bench.py --batch 8`"}
        del coach8
        coach = None
    if world == 1 and not strong and args.workload == 2 and args.size == 1024 and args.conv_precision == "f32" and not args.no_config3:
        # BASELINE configs[2] in the same run, so that a driver-run line exists for it: region-attention mask (the real one) +
        # clip_loss + id_loss (IR-SE50 on the conv engine), batch 8.  Reported beside the headline, never as `value`.
        coach = None
        torch.cuda.empty_cache()
        b3 = 8
        coach3 = build_coach(args.size, b3, device, False, 'hip', 3)
        w3 = synthetic_latents(coach3.net.decoder, b3, rank)
        mask3 = make_mask(coach3, b3, args.size, rank, device)
        step3, graph3 = (lambda: coach3.train_step(w3, mask3)), False
        if use_graph:  # (the headline ran as a hipGraph: so does this step, the mask branch captured between the two passes)
            try:
                g3 = coach3.capture_step(w3, mask3)
                step3, graph3 = (lambda: g3(w3, mask3)), True
            except Exception:  # noqa: BLE001
                torch.cuda.synchronize()
        n3, ok3 = stabilise(step3)
        barrier()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            last3 = step3()
        barrier()
        dt3 = time.perf_counter() - t3
        out["config3"] = {"value": b3 * args.steps / dt3, "unit": "images/s", "ms_per_step": 1e3 * dt3 / args.steps, "batch": b3,
                          "final_loss": float(last3["loss"]), "stabilise_steps": n3, "hip_graph": graph3,
                          "mask_mean": float(mask3.last.mean()) if hasattr(mask3, "last") else None,
                          "workload": "BASELINE configs[2]: FFHQ-1024 mapper step with the region-attention mask (cluster-pooled, "
                                      "thresholded, blurred; run_attention.py:754-884) blended at layer 13 + clip_loss + id_loss "
                                      "(IR-SE50), batch 8, 1 GPU; same as `bench.py --workload 3 --batch 8`"}
    if world == 1 and not strong and args.workload == 2 and args.size == 1024 and args.conv_precision == "f32" and not args.no_config5:
        # BASELINE configs[4] (invert-and-edit inference) at ONE GPU's share of its 32 images over 8 GPUs, in the same run so that a
        # driver-run figure exists for it.  Reported beside the headline, never as `value`.
        coach = coach3 = None
        torch.cuda.empty_cache()
        out["config5"] = measure_config5(args, rank, world, device, 4)
        out["config5"]["same_as"] = "`bench.py --workload 5 --batch 4`"
        torch.cuda.empty_cache()
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size)
    print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
