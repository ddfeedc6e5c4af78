"""N4: on-disk formats (CPU) -- latent datasets, DDP-prefixed mapper checkpoints, the k-means pickle, Coach checkpoints --
round trips and schema checks against what the reference reads and writes."""
import pickle
import types

import torch

import seeded


def test_latents_datasets_match_reference_semantics():
    from where2edit_amd.latents_dataset import (STYLESPACE_DIMENSIONS, LatentsDataset, StyleSpaceLatentsDataset,
                                                aggregate_loss_dict, convert_s_tensor_to_list)
    w = seeded.tensor("fmt.w", (5, 18, 512))
    ds = LatentsDataset(w, None)
    assert len(ds) == 5 and torch.equal(ds[3], w[3])
    codes = [seeded.tensor(f"fmt.s{i}", (5, 1, c, 1, 1)) for i, c in enumerate(STYLESPACE_DIMENSIONS)]
    sds = StyleSpaceLatentsDataset(codes, None)
    assert len(sds) == 5 and tuple(sds[0].shape) == (1, 26 * 512, 1, 1)
    back = convert_s_tensor_to_list(torch.stack([sds[i] for i in range(5)]))  # a collated batch [B,1,26*512,1,1]
    assert all(torch.equal(a, b) for a, b in zip(back, codes))
    assert sds.latents[:, :, 512 * 25 + 32: 512 * 26].abs().sum() == 0  # zero padding of the 32-channel code
    assert aggregate_loss_dict([{"a": 1.0, "b": 2.0}, {"a": 3.0}]) == {"a": 2.0, "b": 2.0}


def test_latents_dataset_agrees_with_reference_class_on_golden_shapes(tmp_path):
    """torch.save / load_latents round trip in both modes (coach.py:195-221)."""
    from where2edit_amd.latents_dataset import STYLESPACE_DIMENSIONS, load_latents
    w = seeded.tensor("fmt.w2", (3, 18, 512))
    torch.save(w, tmp_path / "w.pt")
    assert torch.equal(load_latents(tmp_path / "w.pt").latents, w)
    codes = [seeded.tensor(f"fmt.t{i}", (3, 1, c, 1, 1)) for i, c in enumerate(STYLESPACE_DIMENSIONS)]
    torch.save(codes, tmp_path / "s.pt")
    assert tuple(load_latents(tmp_path / "s.pt", work_in_stylespace=True).latents.shape) == (3, 1, 26 * 512, 1, 1)


def test_ddp_prefixed_mapper_checkpoint_round_trip(tmp_path):
    from where2edit_amd import checkpoints as ck
    from where2edit_amd.run_attention import FullSpaceMapperFEATClusterLinStyle_Net as Net
    a = Net(14, 1024, 512, attention_layer=7, channel_multiplier=2, cluster_layer=7, clusters=6, cluster_dim=576)
    ck.save_mapper(a, tmp_path / "m.pt")                      # what run_attention.py:1437 writes (DDP: module.*)
    raw = torch.load(tmp_path / "m.pt")
    assert all(k.startswith("module.") for k in raw) and len(raw) == len(a.state_dict())
    b = Net(14, 1024, 512, attention_layer=7, channel_multiplier=2, cluster_layer=7, clusters=6, cluster_dim=576)
    res = ck.load_mapper(b, tmp_path / "m.pt", strict=True)   # try_demo.py:38-42 strips the prefix
    assert not res.missing_keys and not res.unexpected_keys
    assert all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items())
    ck.save_mapper(a, tmp_path / "bare.pt", ddp_prefix=False)
    ck.load_mapper(b, tmp_path / "bare.pt", strict=True)


def test_kmeans_pickle_round_trip(tmp_path):
    from where2edit_amd import checkpoints as ck
    centres = seeded.tensor("fmt.centres", (20, 576)).double()  # sklearn's cluster_centers_ are float64
    ck.save_clusters(centres, tmp_path / "k.pkl")
    with open(tmp_path / "k.pkl", "rb") as f:
        raw = pickle.load(f)                                  # run_attention.py:996-998 reads it exactly like this
    assert torch.is_tensor(raw) and raw.dtype == torch.float64 and torch.equal(raw, centres)
    assert torch.equal(ck.load_clusters(tmp_path / "k.pkl"), centres)


def test_coach_and_generator_checkpoint_schemas(tmp_path):
    from where2edit_amd import checkpoints as ck
    from where2edit_amd.styleclip_mapper import StyleCLIPMapper
    from where2edit_amd.stylegan2 import Generator
    opts = types.SimpleNamespace(no_coarse_mapper=False, no_medium_mapper=False, no_fine_mapper=False, work_in_stylespace=False,
                                 mapper_type="LevelsMapper", stylegan_size=16, checkpoint_path=None, stylegan_weights=None)
    net = StyleCLIPMapper(opts)
    ck.save_coach_checkpoint(net, opts, tmp_path / "c.pt")
    sd, o = ck.load_coach_checkpoint(tmp_path / "c.pt")
    assert o["mapper_type"] == "LevelsMapper" and any(k.startswith("mapper.") for k in sd) and any(k.startswith("decoder.") for k in sd)
    g = Generator(16, 512, 8)
    torch.save({"g_ema": seeded.generator_state_dict(16)}, tmp_path / "g.pt")
    res = ck.load_generator_weights(g, tmp_path / "g.pt")
    assert not res.missing_keys and not res.unexpected_keys
