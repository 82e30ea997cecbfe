"""Datasets for train_ocr.py.  Same sample content as the reference's utils/datasets.py:8-27 ("obss", optional "masks" [K+1,H,W,1] with
the background last).  Reads the reference's HDF5 when h5py and the file are available, otherwise generates random-N5C4S4S2-style
scenes deterministically per index.  With raw_uint8 (the default of get_dataloaders) a sample carries the image as the dataset stores
it — "obss_u8", uint8 HWC — and the permute + /255 of utils/datasets.py:17 runs on the GPU (ocrl_obs_u8_to_f32) after a 4x smaller
upload; raw_uint8=False gives the reference's float CHW "obss"."""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .data import random_sprite_scenes


class SyntheticScenes(Dataset):
    def __init__(self, n, size, seed=0, with_masks=False, num_objs=5, raw_uint8=False):
        self.n, self.size, self.seed, self.with_masks, self.num_objs = int(n), int(size), int(seed), bool(with_masks), num_objs
        self.raw_uint8 = bool(raw_uint8)

    def _obs(self, img):
        if self.raw_uint8:
            return {"obss_u8": torch.from_numpy(img)}
        return {"obss": torch.from_numpy(img).permute(2, 0, 1).float() / 255.0}

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        s = (self.seed * 1000003 + index) & 0x7FFFFFFF
        if self.with_masks:
            img, m = random_sprite_scenes(1, self.size, seed=s, num_objs=self.num_objs, with_masks=True)
            return {**self._obs(img[0]), "masks": torch.from_numpy(m[0])}
        img = random_sprite_scenes(1, self.size, seed=s, num_objs=self.num_objs)
        return self._obs(img[0])


class H5DataSet(Dataset):
    """utils/datasets.py:8-27"""

    def __init__(self, data, raw_uint8=False):
        self._data = data
        self._num_samples = data["obss"].shape[0]
        # the raw path uploads the stored bytes; it is taken only when the file really stores uint8 (the reference's
        # torch.Tensor(x) / 255 accepts any dtype, a blind uint8 cast would wrap or truncate other storage types)
        self._raw_uint8 = bool(raw_uint8) and np.dtype(data["obss"].dtype) == np.uint8

    def __getitem__(self, index):
        res = {}
        for key in self._data.keys():
            if key == "obss" and self._raw_uint8:
                res["obss_u8"] = torch.from_numpy(np.ascontiguousarray(self._data[key][index]))
            elif key == "obss":
                res[key] = torch.Tensor(self._data[key][index]).permute(2, 0, 1) / 255.0
            elif key == "labels":
                res[key] = torch.LongTensor([self._data[key][index]])
            elif key != "num_objs":
                res[key] = torch.Tensor(self._data[key][index])
        return res

    def __len__(self):
        return self._num_samples


def get_dataloaders(config, batch_size, num_workers, rank=0, world=1, seed=0, raw_uint8=True):
    """utils/tools.py:155-178 without the wandb download path"""
    datafile = config.datadir if config.get("datadir") else None
    if datafile and os.path.isfile(datafile):
        try:
            import h5py
        except ImportError as e:
            raise RuntimeError(f"{datafile} exists but h5py is not installed") from e
        f = h5py.File(datafile, "r")
        train, val = H5DataSet(f["TrainingSet"], raw_uint8), H5DataSet(f["ValidationSet"], raw_uint8)
    else:
        wm = bool(config.get("with_masks", False))
        train = SyntheticScenes(config.get("synthetic_train", 100000), config.obs_size, seed=seed * 2 + 1, with_masks=wm, raw_uint8=raw_uint8)
        val = SyntheticScenes(config.get("synthetic_val", 1000), config.obs_size, seed=seed * 2 + 2, with_masks=wm, raw_uint8=raw_uint8)
    sampler = None
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(train, num_replicas=world, rank=rank, shuffle=True, seed=seed, drop_last=True)
    train_dl = DataLoader(train, batch_size, num_workers=num_workers, shuffle=(sampler is None), sampler=sampler, drop_last=True)
    val_dl = DataLoader(val, batch_size)
    return train_dl, val_dl
