"""GeometryLatentDataset (drop-in for denoiser/dataset/dataset.py): reads the reference's pc_data / matching_data
npz files and produces the per-puzzle dict the Denoiser / AutoAgglomerative consume.

Two augmentation paths:
  * device_augment=False (default): the reference's CPU augmentation per sample — random rotation of the whole
    assembly, recentring on the reference part, per-part recentring + random rotation, max-abs normalisation
    (dataset.py:165-215) — with scipy rotations drawn in the same order from numpy's global RNG.
  * device_augment=True: __getitem__ returns the stored geometry only (padded part_pcs_gt) and
    pfpp_hip.augment.augment_batch does the same arithmetic for the whole batch in one HIP kernel on the GPU
    (the 10 DataLoader workers doing scipy rotations per sample are the bottleneck once the step takes 13 ms)."""
from __future__ import annotations

import copy
import os
from typing import Dict

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from pfpp_hip import io as pfio
from pfpp_hip.scheduler import PiecewiseScheduler


class GeometryLatentDataset(Dataset):
    def __init__(self, cfg, data_dir, overfit, data_fn, device_augment: bool = False):
        self.cfg = cfg
        self.mode = data_fn
        self.data_dir = data_dir
        self.device_augment = device_augment
        self.max_num_part = cfg.data.max_num_part
        self.noise_scheduler = PiecewiseScheduler()
        files = sorted(f for f in os.listdir(data_dir) if f.endswith(".npz"))
        if overfit != -1:
            files = files[:overfit]
        self.data_files = files
        matching_dir = cfg.data.matching_data_path if self.mode == "test" else None
        self.data_list = []
        for name in files:
            sample = pfio.load_pc_data(os.path.join(data_dir, name))
            sample.pop("category", None)
            if matching_dir is not None:
                mpath = os.path.join(matching_dir, f"{sample['data_id']}.npz")
                if not os.path.exists(mpath):
                    continue                                     # puzzles without matching data are skipped (dataset.py:57-58)
                sample.update(pfio.load_matching_data(mpath))
            self.data_list.append(sample)

    def __len__(self):
        return len(self.data_list)

    # ------------------------------------------------------------------ geometry helpers (dataset.py:86-163)
    @staticmethod
    def _random_rotation():
        from scipy.spatial.transform import Rotation as R

        rot = R.random().as_matrix()
        quat = R.from_matrix(rot.T).as_quat()[[3, 0, 1, 2]]       # scalar-first quaternion of the inverse rotation
        return rot, quat

    def _pad(self, arr: np.ndarray) -> np.ndarray:
        arr = np.asarray(arr)
        out = np.zeros((self.max_num_part,) + arr.shape[1:], dtype=np.float32)
        out[: arr.shape[0]] = arr
        return out

    @staticmethod
    def _to_initial_frame(pcs, n_pcs, num_parts, trans, quats):
        """by-area points of the assembled shape -> each part's own (recentred, rotated) frame (dataset.py:93-109)"""
        from scipy.spatial.transform import Rotation as R

        out, off = [], 0
        for i in range(num_parts):
            c = pcs[off: off + n_pcs[i]] - trans[i]
            out.append(R.from_quat(quats[i][[1, 2, 3, 0]]).inv().apply(c))
            off += n_pcs[i]
        return np.concatenate(out, axis=0)

    def __getitem__(self, idx) -> Dict[str, object]:
        d = copy.deepcopy(self.data_list[idx])
        num_parts = d["num_parts"]
        gt = d["part_pcs_gt"]
        if self.device_augment:
            d["part_pcs_gt"] = self._pad(gt).astype(np.float32)
            return d
        from scipy.spatial.transform import Rotation as R

        P, N, _ = gt.shape
        rot_g, pose_gt_r = self._random_rotation()               # whole assembly (dataset.py:134-146)
        pts = (rot_g @ gt.reshape(-1, 3).T).T.reshape(P, N, 3)
        pose_gt_t = np.mean(pts[int(np.where(d["ref_part"])[0].item())], axis=0)
        pts = pts - pose_gt_t
        cur_pts, cur_quat, cur_trans = [], [], []
        for i in range(num_parts):
            centroid = np.mean(pts[i], axis=0)
            rot_p, quat = self._random_rotation()
            cur_pts.append((rot_p @ (pts[i] - centroid[None]).T).T)
            cur_quat.append(quat)
            cur_trans.append(centroid)
        cur_pts = self._pad(np.stack(cur_pts, 0)).astype(np.float32)
        cur_quat = self._pad(np.stack(cur_quat, 0)).astype(np.float32)
        cur_trans = self._pad(np.stack(cur_trans, 0)).astype(np.float32)
        if self.mode == "test":
            anchored = R.from_quat(pose_gt_r[[1, 2, 3, 0]]).inv().apply(d["gt_pc_by_area"]) - pose_gt_t     # dataset.py:86-91
            d["part_pcs_by_area"] = self._to_initial_frame(anchored, d["n_pcs"], num_parts, cur_trans, cur_quat).astype(np.float32)
        scale = np.max(np.abs(cur_pts), axis=(1, 2), keepdims=True)
        scale[scale == 0] = 1
        d["part_pcs"] = cur_pts / scale
        d["part_pcs_gt"] = self._pad(gt).astype(np.float32)
        d["part_rots"] = cur_quat
        d["part_trans"] = cur_trans
        d["part_scale"] = scale.squeeze(-1)
        d["init_pose_r"] = pose_gt_r
        d["init_pose_t"] = pose_gt_t
        if getattr(self.cfg.model, "multiple_ref_parts", False) and self.mode == "train":
            self._extra_reference_parts(d)
        return d

    def _extra_reference_parts(self, d) -> None:
        """half of the time promote random neighbours of the reference part to (slightly perturbed) reference parts
        (dataset.py:229-271)"""
        if d["num_parts"] == 2 or np.random.rand() < 0.5:
            return
        ref_part, graph, scale = d["ref_part"], d["graph"], d["part_scale"]
        connected = np.where(graph[np.where(ref_part)[0], :])[1]
        if not [p for p in connected if scale[p] > 0.05]:
            return
        n_large = len([p for p in connected if scale[p] > 0.05])
        chosen = np.random.choice(connected, np.random.randint(0, n_large), replace=False)
        ref_part[chosen] = True
        d["ref_part"] = ref_part
        noise_t, noise_r = torch.randn(d["part_trans"][chosen].shape), torch.randn(d["part_rots"][chosen].shape)
        t = int(torch.randint(0, 50, (1,)))
        ac = self.noise_scheduler.alphas_cumprod[t]              # host tables only: DataLoader workers never touch the GPU library
        sa, sb = float(ac ** 0.5), float((1 - ac) ** 0.5)
        for key, noise in (("part_trans", noise_t), ("part_rots", noise_r)):
            d[key][chosen] = (sa * torch.tensor(d[key][chosen]) + sb * noise).numpy()


def _loader(ds, batch_size, shuffle, drop_last, num_workers):
    return DataLoader(dataset=ds, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, pin_memory=True,
                      drop_last=drop_last, persistent_workers=(num_workers > 0))


def build_geometry_dataloader(cfg, device_augment: bool = False):
    """train / val loaders (dataset.py:276-308)"""
    train = GeometryLatentDataset(cfg, cfg.data.data_dir, cfg.data.overfit, "train", device_augment)
    val = GeometryLatentDataset(cfg, cfg.data.data_val_dir, cfg.data.overfit, "val", device_augment)
    return (_loader(train, cfg.data.batch_size, True, True, cfg.data.num_workers),
            _loader(val, cfg.data.val_batch_size, False, False, cfg.data.num_workers))


def build_test_dataloader(cfg):
    """test loader with the matching data attached (dataset.py:311-331)"""
    ds = GeometryLatentDataset(cfg, cfg.data.data_val_dir, cfg.data.overfit, "test")
    return _loader(ds, cfg.data.val_batch_size, False, False, cfg.data.num_workers)
