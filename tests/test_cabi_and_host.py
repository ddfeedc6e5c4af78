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


def test_driver_build_hook_checks_hold_for_the_built_library(lib):
    """__graft_entry__.build()'s post-build checks (it asserted ABI version 1 two rounds after the ABI moved on: the
    compile passed and the hook raised).  The version now comes from include/w2e.h everywhere."""
    import __graft_entry__ as G
    from where2edit_amd import _lib
    G.check_built_library(lib)
    assert lib.w2e_version() == _lib.header_version() >= 3


def test_version_and_argument_errors_do_not_need_a_gpu(lib):
    from where2edit_amd import _lib
    assert lib.w2e_version() == _lib.header_version()
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
    # w2e_gemm_pk: an output pitch below n would make rows of a slab overlap -- refused before any launch
    lib.w2e_gemm_pk.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
    rc = lib.w2e_gemm_pk(ctypes.c_void_p(64), ctypes.c_void_p(64), ctypes.c_void_p(64), 32, 128, 64, 32, 128, 96, 1, None)
    assert rc != 0 and b"ldc 96 < n 128" in lib.w2e_last_error()
    lib.w2e_upfirdn2d.argtypes = None


def test_a_caller_with_another_planar_pitch_is_refused_not_read_past(lib):
    """Round 3's e7 memory fault: the CALLER allocates the phase-planar transposed-conv output, the kernels index it with
    W2E_PLANAR_PITCH; a caller still on ABI 2's pitch (W+1 rounded to 4 floats) handed over a buffer ~15 % too small.  Every entry point
    that takes a planar tensor now takes the pitch the caller allocated with and refuses anything but its own -- before any HIP call,
    so this needs no GPU (the pointers are never dereferenced)."""
    P, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    lib.w2e_last_error.restype = ctypes.c_char_p
    w = 512
    mine, old = ((w + 1) + 15) & ~15, ((w + 1) + 3) & ~3
    assert mine == 528 and old == 516
    d = P(4096)  # a non-null, aligned dummy address
    lib.w2e_modconv3x3.argtypes = [I, P, P, P, P, P, I, I, I, I, I, I, I, P, P, P, P, P, P]
    rc = lib.w2e_modconv3x3(1, d, d, None, None, d, 1, 64, 32, w, w, old, 0, None, None, None, None, None, None)
    assert rc != 0 and b"row pitch of 516" in lib.w2e_last_error() and b"528" in lib.w2e_last_error()
    lib.w2e_upfirdn2d.argtypes = [P, P, P, L] + [I] * 14 + [P] * 4 + [I, F, F, P]
    rc = lib.w2e_upfirdn2d(d, d, d, 32, 2 * w + 1, 2 * w + 1, 2 * w, 2 * w, 4, 4, 1, 1, 1, 1, 1, 1, old, 0, None, None, None, None, 1, 0.2, 1.0, None)
    assert rc != 0 and b"row pitch of 516" in lib.w2e_last_error()
    lib.w2e_conv3x3.argtypes = [I, P, P, P, P, P, I, I, I, I, I, I, I, P, P, P]
    rc = lib.w2e_conv3x3(1, d, d, None, None, d, 1, 64, 64, 56, 56, ((56 + 1) + 3) & ~3, 0, None, None, None)
    assert rc != 0 and b"row pitch of 60" in lib.w2e_last_error() and b"64" in lib.w2e_last_error()
    lib.w2e_affine_act_bwd.argtypes = [P, P, P, P, P, I, I, I, I, I, P]
    rc = lib.w2e_affine_act_bwd(d, d, None, None, d, 1, 64, 112, 112, 60, None)
    assert rc != 0 and b"row pitch of 60" in lib.w2e_last_error()
    lib.w2e_shortcut_add_bwd.argtypes = [P, P, I, I, I, I, I, I, P]
    rc = lib.w2e_shortcut_add_bwd(d, d, 1, 64, 112, 112, 1, 60, None)
    assert rc != 0 and b"row pitch of 60" in lib.w2e_last_error()
    for fn in (lib.w2e_modconv3x3, lib.w2e_upfirdn2d, lib.w2e_conv3x3, lib.w2e_affine_act_bwd, lib.w2e_shortcut_add_bwd):
        fn.argtypes = None


def test_winograd_selection_is_host_logic_and_respects_the_4gb_descriptor(lib, monkeypatch):
    """the fused kernel addresses x through ONE buffer descriptor (< 4 GB).  At total batch 32 the 32 @ 1024^2
    layer is exactly 4 GiB: the selection must fall back (here: to the direct kernel) instead of handing w2e_wino_fused a tensor it
    refuses; likewise 64 @ 512^2 at 64 and 128 @ 256^2 at 128 (there the GEMM form takes over).  No GPU: shapes only."""
    from where2edit_amd import _lib, functional as K
    monkeypatch.setattr(_lib, "get_option", lambda name: {"conv_precision": 0, "deterministic": 0, "tune_cfg": -1}[name])
    monkeypatch.setattr(K, "WINOGRAD", "auto")

    class X:  # _wino_form only reads the batch
        def __init__(self, b):
            self.shape = (b,)
    assert K._wino_form(X(8), 32, 32, 1024, 1024, None) == K.FUSED
    assert K._wino_form(X(31), 32, 32, 1024, 1024, None) == K.FUSED
    assert K._wino_form(X(32), 32, 32, 1024, 1024, None) == 0          # 4 GiB: past the descriptor -> direct kernel
    assert K._wino_form(X(64), 64, 64, 512, 512, None) == 0
    assert K._wino_form(X(127), 128, 128, 256, 256, None) == K.FUSED
    assert K._wino_form(X(128), 128, 128, 256, 256, None) == 0          # (the GEMM form's V would be 9.7 GB here: past ITS descriptor too)
    assert K._wino_form(X(8), 512, 512, 64, 64, None) == 4 and K._wino_form(X(8), 256, 256, 128, 128, X(8)) == 4
    assert K._wino_form(X(8), 512, 512, 8, 8, None) == 0 and K._wino_form(X(8), 512, 256, 64, 64, None) == 4
    assert not K._gemm_shape_ok(8, 512, 48, 64, 64) and not K._gemm_shape_ok(8, 20, 64, 64, 64) and not K._gemm_shape_ok(8, 64, 64, 30, 32)
    assert K._gemm_shape_ok(16, 128, 256, 28, 28, dot=False) and not K._gemm_shape_ok(16, 128, 256, 28, 28, dot=True)  # 49 tiles: no fused dot
    # the plan entry point is host code too: padded tiles, K split for the layers that would leave CUs idle, workspace size
    tp, sp, ws = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
    lib.w2e_wino_gemm_plan.argtypes = None
    assert lib.w2e_wino_gemm_plan(8, 512, 512, 64, 64, ctypes.byref(tp), ctypes.byref(sp), ctypes.byref(ws)) == 0
    assert (tp.value, sp.value, ws.value) == (2048, 1, 8 * 512 * 8)
    assert lib.w2e_wino_gemm_plan(3, 64, 64, 28, 28, ctypes.byref(tp), ctypes.byref(sp), ctypes.byref(ws)) == 0
    assert tp.value == 160 and sp.value >= 1 and ws.value == 3 * 64 * 1 + (sp.value * 3 * 64 * 28 * 28 if sp.value > 1 else 0)
    assert lib.w2e_wino_gemm_plan(8, 512, 48, 64, 64, ctypes.byref(tp), ctypes.byref(sp), ctypes.byref(ws)) != 0  # N % 64
    with pytest.raises(ValueError):
        K._parse_winograd("off")  # (an unknown value used to mean "auto")
    with pytest.raises(ValueError):
        K._parse_winograd("f2")   # (the F(2x2,3x3) form went with the vendor GEMM)
    assert K._parse_winograd("") == "auto" and K._parse_winograd("0") is False and K._parse_winograd("4") == 4
    with pytest.raises(ValueError):
        K.set_winograd(2)


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
    # built the reference's way (nothing frozen by the caller): the wrapper freezes exactly the conv weights the kernels cannot
    # differentiate; every other decoder parameter keeps requires_grad (the reference leaves all of them trainable, coach.py:91)
    from where2edit_amd.stylegan2 import ModulatedConv2d
    conv_w = {id(m.weight) for m in net.decoder.modules() if isinstance(m, ModulatedConv2d)}
    assert conv_w and all(p.requires_grad == (id(p) not in conv_w) for p in net.decoder.parameters())
    assert all(p.requires_grad for p in net.mapper.parameters())
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


def test_trainable_conv_weight_is_refused_and_nothing_computes_on_the_cpu():
    """A 3x3 weight that requires grad (decoder fine-tuning, off this path) is refused with an error naming the fix -- there is no stock-op
    backend behind ModulatedConv2d -- and with frozen conv weights a CPU tensor is refused by the kernels' door: no CPU path either."""
    from where2edit_amd.stylegan2 import ModulatedConv2d, StyledConv, freeze_conv_weights
    x, w = torch.randn(1, 8, 4, 4), torch.randn(1, 512)
    for m in (ModulatedConv2d(8, 8, 3, 512), StyledConv(8, 8, 3, 512)):
        with pytest.raises(RuntimeError, match="freeze_conv_weights"):
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


def test_fused_winograd_kernel_keeps_its_hand_counted_waits():
    """The fused Winograd kernel's transform waves stage their patches with inline-asm `buffer_load_dwordx4 ... lds` and wait for them
    with HAND-WRITTEN `s_waitcnt vmcnt(6)` (csrc/winograd.hip: six DMA instructions per wave and chunk, the newest chunk may fly).
    That is right only while (i) the compiler issues no VMEM load of its own inside those waves' tick loop (its wait would be counted
    on the same in-order counter) and (ii) it merges no vmcnt into the waits it inserts there.  A compiler / ROCm change that breaks
    either would still pass most numeric tests by luck of timing -- so the ISA is checked at build time: every kernel variant is
    compiled to assembly here and the transform-wave region (from the first LDS-DMA instruction to the end of the kernel) must
    contain, for the variants without the fused dot, no other VMEM load at all and no vmcnt wait except the two the source writes."""
    import re
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "where2edit_amd", "csrc", "winograd.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "wino.s")
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                            "--cuda-device-only", "-o", out, src], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        text = open(out).read()
    kernels = re.findall(r"^(_ZN3w2e19wino4_fused3_kernelILi(\d)ELb(\d)ELi(\d+)ELi(\d)EE\w+):[^\n]*\n(.*?)^\s*\.size\s+\1,", text, flags=re.S | re.M)
    assert len(kernels) == 16, [k[0] for k in kernels]  # 4 epilogues x 2 block shapes x 4 / 8 matrix waves
    for name, act, dot, txn, mw, body in kernels:
        lines = [ln.strip() for ln in body.splitlines()]
        first = next(i for i, ln in enumerate(lines) if re.match(r"buffer_load_dwordx4 .* lds$", ln))
        region = lines[first:]
        dma = [ln for ln in region if re.match(r"buffer_load_dwordx4 .* lds$", ln)]
        assert dma and len(dma) % 6 == 0, (name, len(dma))  # whole groups of six per issue site
        assert not any(re.match(r"buffer_load_dwordx4 .* lds$", ln) for ln in lines[:first])  # (the matrix waves issue none)
        region = [ln.split(";")[0].strip() for ln in region]
        waits = [ln for ln in region if ln.startswith("s_waitcnt") and "vmcnt" in ln]
        assert "s_waitcnt vmcnt(6)" in waits, (name, waits)
        if dot == "0":
            other_loads = [ln for ln in region if re.match(r"(global_load|buffer_load|flat_load|scratch_load)", ln) and not ln.endswith(" lds")]
            assert not other_loads, (name, other_loads[:4])
            assert set(waits) <= {"s_waitcnt vmcnt(6)", "s_waitcnt vmcnt(0)"}, (name, sorted(set(waits)))
        # THE invariant of the hand-counted waits, for every variant: between the six DMA instructions of an issue site and the wait the
        # source writes behind them, the wave issues no other VMEM operation -- no load, no store, no spill (a spill is a scratch access
        # on the same in-order counter: `vmcnt(6)` would then leave one DMA of the OLDER chunk in flight).  Elsewhere in these waves a
        # spill only costs time (the 64-channel fused-dot variant reloads one register once, right after the prologue's barrier).
        vmem = re.compile(r"(buffer_|global_|flat_|scratch_)")
        i = 0
        while i < len(region):
            if re.match(r"buffer_load_dwordx4 .* lds$", region[i]):
                group = [j for j in range(i, min(i + 40, len(region))) if re.match(r"buffer_load_dwordx4 .* lds$", region[j])][:6]
                assert len(group) == 6, (name, i)
                j = group[-1] + 1
                while j < len(region) and not region[j].startswith("s_waitcnt vmcnt("):
                    assert not vmem.match(region[j]) or re.match(r"buffer_load_dwordx4 .* lds$", region[j]), (name, region[j])
                    j += 1
                i = group[-1] + 1
            else:
                i += 1
