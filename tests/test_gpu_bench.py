"""bench.py itself on the GPU, as subprocesses: the N > 1 launcher with ranks that share this box's one GPU (gloo), the captures
made while a process group is alive (RCCL at world size 1, gloo at 2), the strong-scaling mode and the fields that make an N > 1
line self-describing.  Short runs (2-3 steps, no stabilisation): what is checked is that the paths run and what the line says,
not a rate."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--steps", "2", "--warmup", "1", "--no-stabilise", "--no-kernel-timing", "--no-cpu-baseline", "--no-preview", "--no-n1-b8",
         "--no-config3", "--no-config5"]


def _bench(*argv, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]
    assert len(lines) == 1 and lines[0].startswith("{"), (lines, r.stderr[-1000:])  # ONE JSON line on stdout (the contract)
    return json.loads(lines[0])


def test_config5_pipeline_is_captured_with_a_process_group_alive():
    """`bench.py --workload 5 --gpus 2` (gloo: both ranks on this GPU; batch 1 each): every rank captures invert_and_edit as a
    hipGraph AFTER init_process_group -- `--graph on` turns a failed capture into a failed run instead of the silent eager fallback
    of `auto`, so a pass means the replayed graph is what was timed."""
    out = _bench("--workload", "5", "--gpus", "2", "--dist-backend", "gloo", "--batch", "1", "--graph", "on", "--steps", "2", "--warmup", "1")
    cfg = out["config"]
    assert out["n_gpus"] == 2 and cfg["hip_graph"] is True and cfg["hip_graph_note"] is None
    assert cfg["global_batch"] == 2 and cfg["per_gpu_batch"] == 1 and out["value"] > 0


def test_rccl_world_1_group_alive_captures_the_step_and_the_pipeline():
    """tools/dp_rccl_selftest.py: an RCCL (nccl backend) process group of world size 1 -- the backend whose watchdog thread queries
    events on its own, which invalidated global-mode captures -- stays alive while Coach.capture_step AND capture_invert_and_edit
    capture and replay (both go through coach.capture_graph, thread-local capture mode)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_rccl_selftest.py"), "256", "2", "pipeline"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "graphed step ok" in r.stdout and "pipeline captured with the group alive and replayed ok" in r.stdout


def test_two_rank_line_names_its_own_one_gpu_reference():
    """An N > 1 line must be readable without another run: per-GPU batch in `config`, and `n1_equal_batch` = the same step at the same
    per-GPU batch without the collective, measured in the same processes, with the efficiency computed from it."""
    out = _bench("--gpus", "2", "--dist-backend", "gloo", "--batch", "2", "--size", "256", *QUICK)
    cfg = out["config"]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and cfg["per_gpu_batch"] == 2 and cfg["global_batch"] == 4
    assert cfg["hip_graph"] is True and cfg["dist_backend"] == "gloo"
    ref = out["n1_equal_batch"]
    assert ref["per_gpu_batch"] == 2 and ref["value"] > 0
    assert abs(out["parallel_efficiency_vs_n1_equal_batch"] - out["value"] / (2 * ref["value"])) < 1e-9


@pytest.mark.parametrize("gpus", [1, 2])
def test_strong_scaling_mode_keeps_the_global_batch(gpus):
    """--scaling strong: the global batch stays at --global-batch; a GPU's shard runs as micro-batches with accumulated gradients
    (here 8 latents as 8 / gpus / 2 micro-batches of 2 per GPU at 256^2)."""
    extra = ["--dist-backend", "gloo"] if gpus > 1 else []
    out = _bench("--gpus", str(gpus), "--scaling", "strong", "--global-batch", "8", "--batch", "2", "--size", "256", *extra, *QUICK)
    cfg = out["config"]
    assert out["scaling"] == "strong" and cfg["global_batch"] == 8 and cfg["per_gpu_batch"] == 8 // gpus
    assert cfg["micro_batch"] == 2 and cfg["micro_batches_per_step"] == 4 // gpus and cfg["hip_graph"] is True
    assert out["value"] > 0 and cfg["final_loss"] == cfg["final_loss"]
