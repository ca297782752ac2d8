"""On-disk formats of the denoise-and-verify pipeline — host-side I/O, numpy only.

  pc_data/<split>/<data_id:05>.npz     generate_pc_data.py:31-41: data_id, part_valids [20], num_parts, mesh_file_path,
                                       graph [20,20], category, part_pcs_gt [Pv,1000,3], ref_part [20]
  matching_data/<data_id>.npz          Jigsaw_matching/model/modules/matching_base_model.py:630-640: edges [E,2] = (idx2, idx1),
                                       correspondence (object array of [M,2]), gt_pcs [5000,3], critical_pcs_idx [5000],
                                       n_pcs [20], n_critical_pcs [20]
  verifier_data/<name>.npz             verifier/dataset/dataset.py:50-58: cls_gt [E], edge_features [E,6], edge_indices [E,2]
  inference/<dir>/<data_id>/           auto_aggl.py:322-357: predict_<acc>.npy [T,Pv,7], gt.npy [Pv,7], init_pose.npy [7],
                                       mesh_file_path.txt
Writers exist so that synthetic puzzles (pfpp_hip.synthetic) can be stored in exactly the layout the reference's
loaders read, and so that AutoAgglomerative's outputs are consumable by the reference's renderer/."""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np

PC_DATA_KEYS = ("data_id", "part_valids", "num_parts", "mesh_file_path", "graph", "category", "part_pcs_gt", "ref_part")
MATCHING_KEYS = ("edges", "correspondence", "gt_pcs", "critical_pcs_idx", "n_pcs", "n_critical_pcs")
VERIFIER_KEYS = ("cls_gt", "edge_features", "edge_indices")


def save_pc_data(directory: str, *, data_id: int, part_valids: np.ndarray, num_parts: int, mesh_file_path: str,
                 graph: np.ndarray, category: str, part_pcs_gt: np.ndarray, ref_part: np.ndarray) -> str:
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{int(data_id):05}.npz")
    np.savez(path, data_id=int(data_id), part_valids=np.asarray(part_valids), num_parts=int(num_parts),
             mesh_file_path=str(mesh_file_path), graph=np.asarray(graph), category=str(category),
             part_pcs_gt=np.asarray(part_pcs_gt), ref_part=np.asarray(ref_part))
    return path


def load_pc_data(path: str) -> Dict[str, object]:
    with np.load(path) as d:
        missing = [k for k in PC_DATA_KEYS if k not in d.files and k != "category"]
        if missing:
            raise KeyError(f"{path}: missing pc_data entries {missing}")
        return {
            "data_id": d["data_id"].item(), "part_valids": d["part_valids"], "num_parts": d["num_parts"].item(),
            "mesh_file_path": d["mesh_file_path"].item(), "graph": d["graph"], "part_pcs_gt": d["part_pcs_gt"],
            "ref_part": d["ref_part"], "category": d["category"].item() if "category" in d.files else "",
        }


def save_matching_data(directory: str, data_id: int, *, edges: np.ndarray, correspondence: Sequence[np.ndarray],
                       gt_pcs: np.ndarray, critical_pcs_idx: np.ndarray, n_pcs: np.ndarray,
                       n_critical_pcs: np.ndarray) -> str:
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{int(data_id)}.npz")
    corr = np.empty(len(correspondence), dtype=object)
    for i, c in enumerate(correspondence):
        corr[i] = np.asarray(c, dtype=np.int64).reshape(-1, 2)
    np.savez(path, edges=np.asarray(edges, dtype=np.int64).reshape(-1, 2), correspondence=corr, gt_pcs=np.asarray(gt_pcs),
             critical_pcs_idx=np.asarray(critical_pcs_idx), n_pcs=np.asarray(n_pcs), n_critical_pcs=np.asarray(n_critical_pcs))
    return path


def load_matching_data(path: str) -> Dict[str, object]:
    """-> edges, correspondences (list of [M,2]), gt_pc_by_area, critical_pcs_idx, n_pcs, n_critical_pcs
    (the unpacking rules of denoiser/dataset/dataset.py:56-79)"""
    with np.load(path, allow_pickle=True) as d:
        corr = d["correspondence"]
        if corr.shape[0] != 1:
            corr_list = corr.tolist() if corr.dtype == object else [corr[i] for i in range(corr.shape[0])]
        else:
            corr_list = [np.asarray(corr[0] if corr.dtype == object else corr.squeeze())]
        return {
            "edges": d["edges"], "correspondences": [np.asarray(c).reshape(-1, 2) for c in corr_list],
            "gt_pc_by_area": d["gt_pcs"], "critical_pcs_idx": d["critical_pcs_idx"], "n_pcs": d["n_pcs"],
            "n_critical_pcs": d["n_critical_pcs"],
        }


def save_verifier_data(directory: str, name: str, *, cls_gt: np.ndarray, edge_features: np.ndarray,
                       edge_indices: np.ndarray) -> str:
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{name}.npz")
    np.savez(path, cls_gt=np.asarray(cls_gt), edge_features=np.asarray(edge_features), edge_indices=np.asarray(edge_indices))
    return path


def save_inference_data(save_dir: str, *, trajectory: np.ndarray, gt: np.ndarray, init_pose: np.ndarray,
                        mesh_file_path: str, acc) -> List[str]:
    """one puzzle's outputs in the layout of AutoAgglomerative._save_inference_data (auto_aggl.py:322-357):
    trajectory [T, Pv, 7] (valid parts only), gt [Pv, 7], init_pose [7] = (t, q)"""
    os.makedirs(save_dir, exist_ok=True)
    files = [os.path.join(save_dir, f"predict_{acc}.npy"), os.path.join(save_dir, "gt.npy"),
             os.path.join(save_dir, "init_pose.npy"), os.path.join(save_dir, "mesh_file_path.txt")]
    np.save(files[0], np.asarray(trajectory))
    np.save(files[1], np.asarray(gt))
    np.save(files[2], np.asarray(init_pose))
    with open(files[3], "w") as fh:
        fh.write(str(mesh_file_path))
    return files
