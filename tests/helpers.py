"""Shared test helpers (CPU and GPU suites)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind == "f" and z[k].ndim > 0 else z[k]) for k in z.files}


def rel_err(a, b):
    """max|a-b| / max|b|: the 'relative fp32 tolerance' BASELINE.json:north_star states (1e-3)."""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


GRAD_ERRORS = []  # (what, max-norm relative error, cosine, tolerance) of every assert_grad_close call: printed at the end of the run (conftest)

GRAD_TOL = 1e-3  # BASELINE.json:north_star's relative fp32 tolerance, applied to end-to-end gradients too
# The comparisons that keep 5e-3, by name, each with the measurement that says why.  All of them are SMALL generators (16^2 / 64^2:
# few pixels per channel, so one LeakyReLU pre-activation within rounding of 0 -- whose slope then flips 0.2 <-> 1 between two fp32
# evaluation orders -- is a visible share of the max-norm); every 1024^2 step comparison is held to GRAD_TOL (they measure <= 3e-4).
KINK_PRONE = {
    # the ORACLE (fp32, CPU) against the REFERENCE's own fp32 gradient in tests/golden/generator16.npz: 4.4e-3 -- two CPU
    # evaluations of the same network differ by more than 1e-3 here (tests/test_oracle_golden.py, printed at the end of a run)
    "blend 5 grad_w": 5e-3,
    # 64^2 generator, 3 latents: 9.2e-4 ... 9.6e-4 measured on the HIP path in rounds 3-4 -- at the edge of 1e-3
    "grad_w at 64^2": 5e-3,
    # the 64^2 step tests: test_gradient_error_is_within_the_fp32_oracles_own_error puts the oracle's OWN fp32 gradient
    # 3.4e-5 ... 1.67e-3 from its float64 gradient on these latent sets (salts 21, 23, 33, 60)
    "mapper gradients at 64^2": 5e-3,
    "mapper gradients through the blend at 64^2": 5e-3,
}


def assert_grad_close(a, b, what="", tol=None, cos_min=0.9999):
    """Gradients THROUGH the generator: max-norm relative error <= GRAD_TOL (1e-3) AND cosine similarity >= 0.9999.
    fp32 autograd of this network carries discrete noise from LeakyReLU kinks (a pre-activation within rounding of 0 flips its
    slope 0.2<->1); the comparisons where that noise is MEASURED above 1e-3 between two fp32 CPU evaluations, or between the
    oracle's fp32 and float64 gradients, are listed by name in KINK_PRONE and keep 5e-3 -- nothing else does (round 5: the
    default came down from 5e-3; the run prints every measured error with the tolerance it was held to)."""
    if tol is None:
        tol = KINK_PRONE.get(what, GRAD_TOL)
    a, b = torch.as_tensor(a).double().cpu().reshape(-1), torch.as_tensor(b).double().cpu().reshape(-1)
    e = rel_err(a, b)
    cos = torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30)
    GRAD_ERRORS.append((what, e, float(cos), tol))
    assert e <= tol and cos >= cos_min, f"{what}: grad rel err {e:.3e} (tol {tol:.0e}), cosine {cos:.6f}"
