"""Write synthetic Breaking-Bad-shaped puzzles in the reference's on-disk formats (pc_data + matching_data [+ verifier_data]):

    python tools/make_synthetic_dataset.py OUT_DIR [--n 8] [--first-id 0]

OUT_DIR/pc_data/{train,val}/<id:05>.npz, OUT_DIR/matching_data/<id>.npz, OUT_DIR/verifier_data/<id>.npz — readable by
puzzlefusion_plusplus.denoiser.dataset.dataset.GeometryLatentDataset / verifier.dataset.dataset.VerifierDataset (and by
the reference's own loaders)."""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))

from pfpp_hip import io as pfio  # noqa: E402
from pfpp_hip import synthetic  # noqa: E402


def assembled(puzzle):
    """part_pcs_gt [Pv,N,3]: every fragment in the assembled frame (R(q) (scale * p) + t)"""
    pv = int(puzzle["num_parts"])
    R = synthetic._quat_to_mat(puzzle["part_rots"][:pv].astype(np.float64))
    pts = puzzle["part_pcs"][:pv].astype(np.float64) * puzzle["part_scale"][:pv, None].astype(np.float64)
    return (np.einsum("pij,pnj->pni", R, pts) + puzzle["part_trans"][:pv, None].astype(np.float64)).astype(np.float32)


def write_puzzle(out: Path, pid: int, split: str, num_points: int = 1000):
    pz = synthetic.make_puzzle(pid, num_points=num_points)
    pv = int(pz["num_parts"])
    gt = assembled(pz)
    cent = gt.mean(1)
    dist = np.linalg.norm(cent[:, None] - cent[None], axis=-1)
    graph = np.zeros((20, 20), dtype=bool)
    graph[:pv, :pv] = (dist < np.sort(dist, axis=1)[:, min(3, pv - 1)][:, None]) & ~np.eye(pv, dtype=bool)
    graph |= graph.T
    pfio.save_pc_data(str(out / "pc_data" / split), data_id=pid, part_valids=pz["part_valids"], num_parts=pv,
                      mesh_file_path=f"synthetic/{pid:05}", graph=graph, category="synthetic", part_pcs_gt=gt,
                      ref_part=pz["ref_part"])
    import torch

    batch = {k: torch.from_numpy(np.asarray(v))[None] for k, v in pz.items()}
    m = synthetic.make_matching(batch, seed=pid)
    # gt_pcs: the by-area points in the ASSEMBLED frame (the dataset moves them into each part's frame itself)
    n_pcs = m["n_pcs"][0].numpy()
    by_area = m["part_pcs_by_area"][0].numpy().astype(np.float64)
    R = synthetic._quat_to_mat(pz["part_rots"][:pv].astype(np.float64))
    off, world = 0, []
    for i in range(pv):
        world.append(by_area[off: off + n_pcs[i]] @ R[i].T + pz["part_trans"][i].astype(np.float64))
        off += n_pcs[i]
    gt_pcs = np.zeros_like(by_area)
    gt_pcs[:off] = np.concatenate(world)
    pfio.save_matching_data(str(out / "matching_data"), pid, edges=m["edges"][0].numpy(), correspondence=m["correspondences"],
                            gt_pcs=gt_pcs.astype(np.float32), critical_pcs_idx=m["critical_pcs_idx"][0].numpy(), n_pcs=n_pcs,
                            n_critical_pcs=m["n_critical_pcs"][0].numpy())
    rng = np.random.default_rng(pid)
    iu = np.stack(np.triu_indices(pv, k=1), -1)
    pfio.save_verifier_data(str(out / "verifier_data"), f"{pid:05}", cls_gt=rng.integers(0, 2, len(iu)),
                            edge_features=rng.integers(0, 60, (len(iu), 6)).astype(np.float32), edge_indices=iu)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--first-id", type=int, default=0)
    ap.add_argument("--points", type=int, default=1000)
    a = ap.parse_args()
    out = Path(a.out)
    for i in range(a.n):
        write_puzzle(out, a.first_id + i, "train" if i < max(1, int(0.75 * a.n)) else "val", a.points)
    print(f"wrote {a.n} puzzles under {out}")


if __name__ == "__main__":
    main()
