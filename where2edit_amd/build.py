"""Builds libw2e.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m where2edit_amd.build [--force]

Cross-compiles without a GPU.  The .so is git-ignored but travels with the gpurun snapshot.
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
# W2E_LIB_PATH: build / load another file than lib/libw2e.so (a -DW2E_TUNING diagnostic build kept beside the shipped library)
LIB_PATH = os.environ.get("W2E_LIB_PATH") or os.path.join(LIB_DIR, "libw2e.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("W2E_HIPCC_FLAGS", "").split()  # e.g. -DW2E_STAMPS: diagnostic build with per-phase cycle stamps


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(PKG, "..", "include", "w2e.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every .hip under csrc/ into lib/libw2e.so.  Raises on any compiler error.  Objects are kept under lib/obj/
    (git-ignored) and a translation unit is recompiled only when it, a header or the flags changed."""
    if not force and not _stale():
        return LIB_PATH
    obj_dir = os.path.join(LIB_DIR, "obj" if not os.environ.get("W2E_LIB_PATH") else "obj_" + os.path.basename(LIB_PATH))
    os.makedirs(obj_dir, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h"))
    stamp = os.path.join(obj_dir, "flags.txt")
    flags_now = " ".join(FLAGS)
    if force or not os.path.exists(stamp) or open(stamp).read() != flags_now:
        for o in glob.glob(os.path.join(obj_dir, "*.o")):
            os.remove(o)
    objs = []
    procs = []
    for src in sources():  # one hipcc per stale translation unit, in parallel
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        if os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in [src] + headers):
            continue
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = None
    for src, obj, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            if os.path.exists(obj):
                os.remove(obj)
            failed = failed or f"hipcc failed on {src}:\n{out}"
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError(failed)
    with open(stamp, "w") as f:
        f.write(flags_now)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    if verbose:
        print(f"built {LIB_PATH} ({len(procs)} of {len(objs)} translation units recompiled)")
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
