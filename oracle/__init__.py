"""CPU oracle for the Where2edit latent-editing hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32, stock
ATen ops, torch autograd) restatement of the reference algorithm for every row
of SURVEY.md section 8(a).  It exists so that the HIP path can be checked
against something that runs without a GPU and without `/root/reference`.

Who may import it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` -- as the checker / the timed CPU baseline, never as the
thing shipped.  Nothing under `where2edit_amd/` imports `oracle`.

Pinning: `tests/golden/*.npz` hold outputs captured by importing the
reference itself in the build container (`tests/golden/make_golden.py`);
`tests/test_oracle_golden.py` checks this restatement against them.  The CLIP
towers are third-party code absent from `/root/reference` (OpenAI `clip`,
pinned `clip=1.0` / openai/CLIP@8a665a68 in the reference's requirements.txt:31
and cog.yaml:22): for them parity with the *reference* is UNPINNED; the
restatement follows the published ViT-B/32 architecture and is cross-checked
against the independent `transformers` CLIP implementation with shared random
weights (fixture `clip_hf_tiny.npz`).

Functional style on purpose: every function takes a rosinality / OpenAI-CLIP
format ``state_dict`` (the checkpoint schema of SURVEY.md section 3.5) so the same
tensors can be loaded into the product modules and compared.
"""
