"""CPU-side checks: the C-ABI library builds, loads and exports every symbol include/*.h declares
(no compute calls -- there is no GPU here), and the host-side mirror of the reference interface
(module names, state_dict schema, opts handling, error behaviour)."""
import ctypes
import glob
import os
import re
import types

import pytest
import torch

import seeded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from where2edit_amd import build
    return ctypes.CDLL(build.build(verbose=False))


def _declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        names += re.findall(r"^\s*(?:int|const char\*|size_t)\s+(w2e_\w+)\s*\(", open(h).read(), flags=re.M)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(lib):
    names = _declared_symbols()
    assert len(names) >= 14
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_prototypes_cover_the_header():
    from where2edit_amd import _lib, _lib_vit, irse_hip, run_attention
    assert sorted(list(_lib._PROTOS) + list(_lib_vit.PROTOS) + list(run_attention.PROTOS) + list(irse_hip.PROTOS)) == _declared_symbols()


def test_version_and_argument_errors_do_not_need_a_gpu(lib):
    assert lib.w2e_version() == 3
    # options: read from the environment once at load, then only through the ABI; unknown names are refused
    assert lib.w2e_set_option(b"deterministic", b"1") == 0
    v = ctypes.c_int(-1)
    assert lib.w2e_get_option(b"deterministic", ctypes.byref(v)) == 0 and v.value == 1
    assert lib.w2e_set_option(b"deterministic", b"0") == 0
    assert lib.w2e_set_option(b"no_such_option", b"1") != 0
    assert lib.w2e_get_option(b"tuning_build", ctypes.byref(v)) == 0 and v.value == 0  # the shipped library cannot skip work
    lib.w2e_last_error.restype = ctypes.c_char_p
    # argument validation happens before any HIP call: a null tensor is refused with a message
    rc = lib.w2e_clip_preproc_fwd(None, None, ctypes.c_int64(1), 1024, None)
    assert rc != 0 and b"null" in lib.w2e_last_error()
    lib.w2e_upfirdn2d.argtypes = None


def test_generator_state_dict_schema_matches_rosinality_checkpoints():
    """171 keys at 1024; the seeded schema was loaded strict=True into the reference."""
    from where2edit_amd.stylegan2 import Generator
    for size in (16, 256):
        g = Generator(size, 512, 8)
        sd = seeded.generator_state_dict(size)
        assert set(g.state_dict()) == set(sd)
        for k, v in g.state_dict().items():
            assert tuple(v.shape) == tuple(sd[k].shape), k
        g.load_state_dict(sd, strict=True)
    from where2edit_amd.attention_model import Generator as AG
    assert set(AG(16, 512, 8).state_dict()) == set(seeded.generator_state_dict(16))


def test_generator_1024_key_count():
    import math
    keys = seeded.generator_state_dict(1024)
    assert len(keys) == 171
    from where2edit_amd.stylegan2 import Generator
    with torch.device("meta"):
        g = Generator(1024, 512, 8)
    assert set(g.state_dict()) == set(keys) and g.n_latent == 18 and g.num_layers == 17


def _opts(**kw):
    base = dict(no_coarse_mapper=False, no_medium_mapper=False, no_fine_mapper=False, work_in_stylespace=False,
                mapper_type="LevelsMapper", stylegan_size=16, checkpoint_path=None, stylegan_weights=None)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_mapper_modules_mirror_reference_keys():
    from where2edit_amd import latent_mappers as lm
    assert lm.STYLESPACE_DIMENSIONS == [512] * 15 + [256] * 3 + [128] * 3 + [64] * 3 + [32] * 2
    m = lm.LevelsMapper(_opts())
    assert set(m.state_dict()) == set(seeded.mapper_state_dict(["course_mapping.", "medium_mapping.", "fine_mapping."]))
    assert sum(p.numel() for p in m.parameters()) == 3 * 4 * (512 * 512 + 512)
    m2 = lm.LevelsMapper(_opts(no_coarse_mapper=True, no_fine_mapper=True))
    assert set(m2.state_dict()) == set(seeded.mapper_state_dict(["medium_mapping."]))
    assert set(lm.SingleMapper(_opts()).state_dict()) == set(seeded.mapper_state_dict(["mapping."]))
    wo = lm.WithoutToRGBStyleSpaceMapper(_opts())
    assert wo.STYLESPACE_INDICES_WITHOUT_TORGB == [i for i in range(26) if i % 3 != 1]
    assert len(list(lm.FullStyleSpaceMapper(_opts()).children())) == 26


def test_styleclip_mapper_surface(tmp_path):
    from where2edit_amd.styleclip_mapper import StyleCLIPMapper, get_keys
    ck = tmp_path / "g.pt"
    torch.save({"g_ema": seeded.generator_state_dict(16)}, ck)
    net = StyleCLIPMapper(_opts(stylegan_weights=str(ck)))
    assert hasattr(net, "mapper") and hasattr(net, "decoder") and hasattr(net, "face_pool") and net.opts.stylegan_size == 16
    assert torch.equal(net.decoder.state_dict()["conv1.conv.weight"], seeded.generator_state_dict(16)["conv1.conv.weight"])
    full = tmp_path / "full.pt"
    torch.save({"state_dict": net.state_dict(), "opts": vars(net.opts)}, full)
    net2 = StyleCLIPMapper(_opts(checkpoint_path=str(full)))
    assert set(get_keys(torch.load(full), "mapper")) == set(net2.mapper.state_dict())
    with pytest.raises(Exception, match="not a valid mapper"):
        StyleCLIPMapper(_opts(mapper_type="Nope"))
    for kw, cls in ((dict(mapper_type="SingleMapper"), "SingleMapper"), (dict(work_in_stylespace=True), "WithoutToRGBStyleSpaceMapper")):
        assert type(StyleCLIPMapper(_opts(**kw)).mapper).__name__ == cls


def test_ops_refuse_cpu_tensors_loudly():
    """No CPU fallback: the product path must fail, not silently compute elsewhere."""
    from where2edit_amd.op import fused_leaky_relu, upfirdn2d
    from where2edit_amd.stylegan2 import Generator
    with pytest.raises(RuntimeError, match="GPU only"):
        upfirdn2d(torch.randn(1, 1, 8, 8), torch.ones(2, 2))
    with pytest.raises(RuntimeError, match="GPU only"):
        fused_leaky_relu(torch.randn(2, 4), torch.zeros(4))
    g = Generator(16, 512, 8)
    with pytest.raises(RuntimeError, match="GPU only"):
        g([torch.randn(1, 512)])


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from where2edit_amd import _lib, build
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.load()


def test_irse50_backbone_matches_reference_fixture():
    """A9: the IR-SE50 ArcFace backbone (stock torch ops) reproduces the reference's Backbone on seeded weights,
    with the reference's state_dict keys.  Runs on CPU: it is not a hand-kernel target."""
    from helpers import assert_close, golden
    from where2edit_amd.id_loss import Backbone, IDLoss
    g = golden("irse")
    net = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se").eval()
    assert sorted(net.state_dict()) == [str(k) for k in g["keys"]]
    net.load_state_dict(seeded.irse_fill(net.state_dict()), strict=True)
    with torch.no_grad():
        y = net(seeded.tensor("irse.x", (2, 3, 112, 112), 0.5))
    assert_close(y, g["feats"], 1e-4)
    loss_mod = IDLoss(types.SimpleNamespace(ir_se50_weights=None))
    a = seeded.tensor("irse.a", (1, 3, 256, 256), 0.5)
    with torch.no_grad():
        l_same, zero = loss_mod(a, a)
    assert abs(float(l_same)) < 1e-5 and zero == 0


def test_trainable_conv_weight_never_computes_on_the_cpu():
    """A 3x3 weight that requires grad (decoder fine-tuning) routes the layer to the per-sample-weight composition on stock GPU ops
    (announced by a warning; the weight gradients are tested against the oracle on the GPU).  Neither that path nor the frozen
    fast path may compute on CPU tensors: both refuse them."""
    import warnings
    from where2edit_amd.stylegan2 import ModulatedConv2d, StyledConv, freeze_conv_weights
    x, w = torch.randn(1, 8, 4, 4), torch.randn(1, 512)
    for m in (ModulatedConv2d(8, 8, 3, 512), StyledConv(8, 8, 3, 512)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with pytest.raises(RuntimeError, match="GPU only"):
                m(x, w) if isinstance(m, ModulatedConv2d) else m(x, w, noise=torch.zeros(1, 1, 4, 4))
        freeze_conv_weights(m)
        assert m.noise.weight.requires_grad if isinstance(m, StyledConv) else m.modulation.weight.requires_grad
        with pytest.raises(RuntimeError, match="GPU only"):  # the kernels refuse CPU tensors
            m(x, w) if isinstance(m, ModulatedConv2d) else m(x, w, noise=torch.zeros(1, 1, 4, 4))


def test_text_feature_cache_is_keyed_on_contents_not_addresses():
    """encode_text_cached must not return a previous prompt's features when a new token tensor lands on a recycled
    address, and must forget everything on load_state_dict."""
    from make_golden import CLIP_TINY as c
    from where2edit_amd.clip_vit import CLIP
    clip = CLIP(embed_dim=c["embed_dim"], vision_layers=c["vision_layers"], vision_width=c["vision_width"],
                context_length=c["context_length"], vocab_size=c["vocab_size"], transformer_width=c["text_width"],
                transformer_heads=1, transformer_layers=c["text_layers"])
    clip.load_state_dict(seeded.clip_state_dict(**c), strict=True)
    n = c["context_length"]
    t1 = torch.zeros(1, n, dtype=torch.int64)
    t1[0, :3] = torch.tensor([5, 6, c["vocab_size"] - 1])
    f1 = clip.encode_text_cached(t1).clone()
    assert clip.encode_text_cached(t1) is clip._text_cache[3]
    t2 = t1.clone()
    t2[0, 1] = 9
    f2 = clip.encode_text_cached(t2)
    assert not torch.equal(f1, f2) and torch.equal(f2, clip.encode_text(t2))
    t2[0, 1] = 6  # in-place edit back to prompt 1: same object, new version
    assert torch.allclose(clip.encode_text_cached(t2), f1)
    clip.load_state_dict(clip.state_dict())
    assert clip._text_cache is None


def test_bench_refuses_tuning_variables():
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=dict(os.environ, W2E_TUNE_SKIP="2"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "W2E_TUNE_SKIP" in (r.stderr + r.stdout)
