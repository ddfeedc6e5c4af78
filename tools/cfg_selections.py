#!/usr/bin/env python3
"""Records which conv tile / split-K factor libw2e.so picks for every w2e_modconv3x3 / w2e_conv3x3 launch of one bench.py
step (the `tune_print` option: one line per launch on stderr).  The selection depends on the batch, so the file lists the
steps the bench and the driver really run: workload 2 at batch 4 (BASELINE configs[1]) and at batch 8 (configs[3]'s per-rank
workload), workload 3 at batch 8 (configs[2]).

    python tools/cfg_selections.py [out.txt]        (default: stdout)
"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from where2edit_amd import _lib  # noqa: E402
from where2edit_amd import functional as K  # noqa: E402


def one_step(workload, batch):
    dev = "cuda:0"
    coach = bench.build_coach(1024, batch, dev, False, "hip", workload)
    w = bench.synthetic_latents(coach.net.decoder, batch, 0)
    mask = bench.make_mask(coach, batch, 1024, 0, dev) if workload == 3 else None
    coach.train_step(w, mask)  # builds the lazily cached packs (their launches are not part of a steady-state step)
    torch.cuda.synchronize()
    sys.stderr.flush()
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        saved = os.dup(2)
        os.dup2(tmp.fileno(), 2)  # the library prints with fprintf(stderr)
        try:
            _lib.set_option("tune_print", 1)
            K.WINO_LOG = []
            coach.train_step(w, mask)
            torch.cuda.synchronize()
        finally:
            _lib.set_option("tune_print", 0)
            wino, K.WINO_LOG = K.WINO_LOG, None
            os.dup2(saved, 2)
            os.close(saved)
        tmp.seek(0)
        lines = [ln for ln in tmp.read().decode().splitlines() if ln.startswith("modconv mode") or ln.startswith("  ")]
    lines += wino  # (the Winograd-form layers are chosen on the Python side: functional._wino_ok)
    del coach
    torch.cuda.empty_cache()
    return lines


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    for workload, batch, what in ((2, 4, "BASELINE configs[1]"), (2, 8, "configs[3] per-rank workload"), (3, 8, "configs[2]")):
        lines = one_step(workload, batch)
        n = sum(ln.startswith("modconv mode") for ln in lines)
        out.write(f"# tile selections of one bench.py --workload {workload} --batch {batch} step ({what}; 1024^2): {n} conv launches\n")
        out.write("\n".join(lines) + "\n")
    if out is not sys.stdout:
        out.close()


if __name__ == "__main__":
    main()
