"""mapper/datasets/latents_dataset.py + mapper/training/train_utils.py surface: the on-disk latent formats of the
StyleCLIP-mapper trainer (`torch.load(latents_train_path)`, coach.py:182-221) and the helpers that move between the
padded S-space tensor and the list of 26 per-layer codes."""
import torch
from torch.utils.data import Dataset

STYLESPACE_DIMENSIONS = [512 for _ in range(15)] + [256, 256, 256] + [128, 128, 128] + [64, 64, 64] + [32, 32]


class LatentsDataset(Dataset):
    """latents_dataset.py:5-16: a [N,18,512] W+ tensor, item = latents[index]."""

    def __init__(self, latents, opts):
        self.latents = latents
        self.opts = opts

    def __len__(self):
        return self.latents.shape[0]

    def __getitem__(self, index):
        return self.latents[index]


class StyleSpaceLatentsDataset(Dataset):
    """latents_dataset.py:18-37: a list of 26 S-space codes [N,1,C_l,1,1]; every code is zero-padded to 512 channels
    and the list is concatenated along dim 2 -> [N,1,26*512,1,1] (convert_s_tensor_to_list undoes it per batch)."""

    def __init__(self, latents, opts):
        padded = []
        for latent in latents:
            latent = latent.cpu()
            if latent.shape[2] != 512:
                latent = torch.cat([latent, torch.zeros((latent.shape[0], 1, 512 - latent.shape[2], 1, 1))], dim=2)
            padded.append(latent)
        self.latents = torch.cat(padded, dim=2)
        self.opts = opts

    def __len__(self):
        return len(self.latents)

    def __getitem__(self, index):
        return self.latents[index]


def convert_s_tensor_to_list(batch):
    """train_utils.py:17-21"""
    return [batch[:, :, 512 * i: 512 * i + STYLESPACE_DIMENSIONS[i]] for i in range(len(STYLESPACE_DIMENSIONS))]


def aggregate_loss_dict(agg_loss_dict):
    """train_utils.py:3-14: mean of every key over a list of loss dicts."""
    vals = {}
    for output in agg_loss_dict:
        for key in output:
            vals.setdefault(key, []).append(output[key])
    return {k: (sum(v) / len(v) if len(v) > 0 else 0) for k, v in vals.items()}


def load_latents(path, opts=None, work_in_stylespace=False):
    """coach.py:195-221: `torch.load` of a W+ tensor, or of the list of S-space codes."""
    data = torch.load(path, map_location="cpu")
    return StyleSpaceLatentsDataset(data, opts) if work_in_stylespace else LatentsDataset(data, opts)
