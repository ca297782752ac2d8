"""VerifierDataset (drop-in for verifier/dataset/dataset.py): per-puzzle edge histograms + labels from the
reference's verifier_data npz files, padded to the 190 possible edges of 20 nodes."""
from __future__ import annotations

import copy
import os

import numpy as np
from torch.utils.data import DataLoader, Dataset

MAX_NODES = 20


class VerifierDataset(Dataset):
    def __init__(self, data_dir, overfit, mode):
        self.max_nodes = MAX_NODES
        self.max_edges = MAX_NODES * (MAX_NODES - 1) // 2
        files = sorted(f for f in os.listdir(data_dir) if f.endswith(".npz"))
        if overfit != -1:
            files = files[:overfit]
        cut = int(0.8 * len(files))                              # first 80 % train, last 20 % val (dataset.py:41-47)
        self.data_files = files[:cut] if mode == "train" else files[cut:] if mode == "val" else files
        self.data_list = []
        for name in self.data_files:
            with np.load(os.path.join(data_dir, name)) as d:
                n = d["edge_indices"].shape[0]
                valid = np.zeros(self.max_edges, dtype=np.float32)
                valid[:n] = 1
                self.data_list.append({
                    "cls_gt": self._pad(d["cls_gt"].astype(np.int64)).astype(np.float32),
                    "edge_features": self._pad(d["edge_features"]).astype(np.float32),
                    "edge_indices": self._pad(d["edge_indices"]).astype(np.int64),
                    "edge_valids": valid, "num_edges": n,
                })

    def _pad(self, arr):
        out = np.zeros((self.max_edges,) + tuple(arr.shape[1:]), dtype=np.float32)
        out[: arr.shape[0]] = arr
        return out

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, index):
        d = copy.deepcopy(self.data_list[index])
        hist = d["edge_features"]
        count = np.sum(hist, axis=1)
        d["edge_features"] = np.concatenate([hist / np.where(count == 0, 1, count)[:, None], count[:, None]], axis=1)
        return d


def build_geometry_dataloader(cfg):
    def loader(mode, batch_size, shuffle, drop_last):
        return DataLoader(dataset=VerifierDataset(cfg.data.verifier_data_path, cfg.data.overfit, mode), batch_size=batch_size,
                          shuffle=shuffle, num_workers=cfg.data.num_workers, pin_memory=True, drop_last=drop_last,
                          persistent_workers=(cfg.data.num_workers > 0))

    return loader("train", cfg.data.batch_size, True, True), loader("val", cfg.data.val_batch_size, False, False)
