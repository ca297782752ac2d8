"""CPU, world_size 2 over gloo: the N > 1 path of the benchmark = disjoint puzzle shards, no data-path
collective, barrier + max-over-ranks clock + summed unit count."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]



def _free_port() -> int:
    """an unused TCP port on 127.0.0.1 (rendezvous of the spawned ranks)"""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

def _run_two_ranks(worker, timeout=120):
    return _run_ranks(worker, 2, timeout)


def _run_ranks(worker, world, timeout=120):
    """spawn `world` ranks of `worker(rank, world, port, queue)` and return what rank 0 put on the queue; one retry on a fresh
    port if the rendezvous fails (another process can grab the port between _free_port() and the bind of the store)"""
    last = None
    for attempt in range(2):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            out = q.get(timeout=timeout)
            for p in procs:
                p.join(timeout=60)
            if all(p.exitcode == 0 for p in procs):
                return out
            last = RuntimeError(f"exit codes {[p.exitcode for p in procs]}")
        except Exception as e:  # queue.Empty: a rank died before reporting
            last = e
        for p in procs:
            if p.is_alive():
                p.terminate()
            p.join(timeout=30)
    raise AssertionError(f"{world}-rank run failed twice: {last!r}")


def _worker(rank, world, port, out):
    for p in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from pfpp_hip import synthetic
    from pfpp_hip.parallel import max_over_ranks, shard_range, sum_over_ranks

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard_range(6, rank, world)
    data = synthetic.make_batch(a, b - a, num_points=64)          # each rank builds only its own puzzles
    frags = float(data["part_valids"].sum())
    dist.barrier()
    clock = max_over_ranks(1.0 + rank)                            # slowest rank defines the time
    total = sum_over_ranks(frags)
    ids = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(ids, torch.tensor([a, b]))
    if rank == 0:
        out.put((clock, total, [t.tolist() for t in ids]))
    dist.destroy_process_group()


def test_two_rank_sharding_and_clock():
    from pfpp_hip import synthetic

    clock, total, ids = _run_two_ranks(_worker)
    assert clock == 2.0
    assert ids == [[0, 3], [3, 6]]
    ref = float(synthetic.make_batch(0, 6, num_points=64)["part_valids"].sum())
    assert total == ref                                            # shards cover every puzzle exactly once


def _grad_worker(rank, world, port, out):
    for p in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from pfpp_hip.parallel import GradExchange

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    ranges = [(100, 400), (400, 700), (700, 900)]                  # three "layers"; [0,100) and [900,1000) = the rest
    grads = torch.arange(n, dtype=torch.float32) * (rank + 1)
    ex = GradExchange(grads, ranges)
    for i in reversed(range(len(ranges))):                          # the backward finishes layers last-to-first
        ex.layer_done(i)
    ex.all_done()
    scale = ex.finish()
    if rank == 0:
        out.put((grads.clone(), scale))
    dist.destroy_process_group()


def test_two_rank_gradient_exchange():
    """the training exchange: every slice of the flat gradient buffer is summed exactly once"""
    grads, scale = _run_two_ranks(_grad_worker)
    assert scale == 0.5
    assert torch.equal(grads, torch.arange(1000, dtype=torch.float32) * 3)   # rank0 (x1) + rank1 (x2)


def _sparse_worker(rank, world, port, out):
    for p in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from pfpp_hip.parallel import GradExchange

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # a 40-row x 4 "embedding table" gradient opens the buffer ([0,160)), then dense rest, one layer [200,300)
    grads = torch.zeros(320)
    grads[160:] = torch.arange(160, dtype=torch.float32) * (rank + 1)
    ex = GradExchange(grads, [(200, 300)], sparse_range=(0, 160))
    idx = torch.tensor([3, 7]) + 10 * rank                          # the rows this rank's batch touched
    rows = torch.full((1, 2, 4), float(rank + 1))                   # [tables, batch, width]
    ex.layer_done(0)
    rows_all, idx_all = ex.gather_rows(rows, idx, dim=1)
    table = grads[:160].view(40, 4)
    table.index_add_(0, idx_all, rows_all[0])                       # every rank scatter-adds ALL ranks' rows
    ex.all_done()
    scale = ex.finish()
    if rank == 0:
        out.put((grads.clone(), idx_all.tolist(), scale))
    dist.destroy_process_group()


def test_two_rank_sparse_row_exchange():
    """embedding-table gradients travel as (rows, indices) instead of a dense all-reduce: the result equals the summed
    dense gradient and the dense slices around the table are still reduced exactly once"""
    grads, idx_all, scale = _run_two_ranks(_sparse_worker)
    assert scale == 0.5 and idx_all == [3, 7, 13, 17]
    want = torch.zeros(320)
    want[160:] = torch.arange(160, dtype=torch.float32) * 3
    t = want[:160].view(40, 4)
    t[3] = t[7] = 1.0
    t[13] = t[17] = 2.0
    assert torch.equal(grads, want)


def test_grad_exchange_is_a_no_op_without_a_process_group():
    from pfpp_hip.parallel import GradExchange

    g = torch.ones(10)
    ex = GradExchange(g, [(2, 5)])
    ex.layer_done(0); ex.all_done()
    assert ex.finish() == 1.0 and torch.equal(g, torch.ones(10))


def _eight_worker(rank, world, port, out):
    for p in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from pfpp_hip.parallel import GradExchange, balanced_assignment, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # uneven "layers" (a gap between the head slice and the first layer, a tail behind the last), a sparse table in front
    n = 4096
    ranges = [(1000, 1700), (1700, 1701), (1701, 3000), (3000, 3900)]
    grads = torch.zeros(n)
    grads[640:] = (torch.arange(n - 640, dtype=torch.float32) % 97) * (rank + 1)
    ex = GradExchange(grads, ranges, sparse_range=(0, 640))
    idx = torch.tensor([rank, 20 + rank, 159 - rank])               # rows of a 160 x 4 table touched by this rank's batch
    rows = torch.full((1, 3, 4), float(rank + 1))
    for i in reversed(range(len(ranges))):
        # two pieces of the head slice (the AdaLN linears of a block) are final with their layer and travel with it
        ex.layer_done(i, extra=((660 + 40 * i, 680 + 40 * i), (900 + 10 * i, 905 + 10 * i)))
    rows_all, idx_all = ex.gather_rows(rows, idx, dim=1)
    grads[:640].view(160, 4).index_add_(0, idx_all, rows_all[0])
    ex.all_done()
    scale = ex.finish()
    a, b = shard_range(37, rank, world)
    spans = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(spans, torch.tensor([a, b]))
    if rank == 0:
        counts = [20, 2, 3, 19, 8, 8, 2, 5, 11, 4, 6, 2, 2, 3, 17, 9]
        out.put((grads.clone(), scale, [t.tolist() for t in spans], balanced_assignment(counts, world),
                 balanced_assignment(counts, world, equal_count=True)))
    dist.destroy_process_group()


def test_eight_rank_gradient_exchange():
    """BASELINE configs[3]'s world size on CPU (gloo): uneven layer slices, a one-element layer, sparse table rows from 8 ranks,
    contiguous puzzle shards that cover 37 puzzles exactly once, fragment-balanced assignment"""
    grads, scale, spans, assign, assign_eq = _run_ranks(_eight_worker, 8, timeout=240)
    assert scale == 0.125
    want = torch.zeros(4096)
    want[640:] = (torch.arange(4096 - 640, dtype=torch.float32) % 97) * 36           # sum of (rank + 1) over 8 ranks
    t = want[:640].view(160, 4)
    for r in range(8):
        for row in (r, 20 + r, 159 - r):
            t[row] += r + 1
    assert torch.equal(grads, want)
    assert spans[0][0] == 0 and spans[-1][1] == 37 and all(spans[i][1] == spans[i + 1][0] for i in range(7))
    assert sorted(sum(assign, [])) == list(range(16))
    counts = [20, 2, 3, 19, 8, 8, 2, 5, 11, 4, 6, 2, 2, 3, 17, 9]
    loads = [sum(counts[i] for i in a) for a in assign]
    assert max(loads) <= 20 and max(loads) - min(loads) <= 8        # no rank carries more than the largest puzzle's worth above the mean
    # fixed per-rank batch size (training): two puzzles each, and still far tighter than dealing them in order (22 .. 26 vs 5 .. 23)
    assert sorted(sum(assign_eq, [])) == list(range(16)) and all(len(a) == 2 for a in assign_eq)
    loads_eq = [sum(counts[i] for i in a) for a in assign_eq]
    in_order = [counts[2 * r] + counts[2 * r + 1] for r in range(8)]
    assert max(loads_eq) - min(loads_eq) < max(in_order) - min(in_order) and max(loads_eq) <= 22


def _accum_worker(rank, world, port, out):
    for p in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from pfpp_hip.parallel import GradExchange

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grads = torch.zeros(100)
    ex = GradExchange(grads, [(40, 80)], sparse_range=(0, 20))
    # micro-batch 1: accumulate locally (what DenoiserTrainEngine.no_sync() sets)
    ex.enabled = False
    grads += float(rank + 1)
    ex.layer_done(0); ex.all_done(dense=True)
    local_after_first = grads.clone()
    # micro-batch 2: the syncing backward reduces the accumulated sum once, the table range densely
    ex.enabled = True
    grads += 10.0 * (rank + 1)
    ex.layer_done(0); ex.all_done(dense=True)
    scale = ex.finish()
    if rank == 0:
        out.put((local_after_first, grads.clone(), scale))
    dist.destroy_process_group()


def test_two_rank_gradient_accumulation_reduces_once():
    """gradient accumulation across ranks: micro-batches before the last one only accumulate (no_sync), the last backward
    all-reduces the accumulated buffer exactly once — including the otherwise row-exchanged table range"""
    first, final, scale = _run_two_ranks(_accum_worker)
    assert torch.equal(first, torch.ones(100))                      # rank 0's own first micro-batch, untouched by rank 1
    assert torch.equal(final, torch.full((100,), 33.0)) and scale == 0.5      # (1 + 10) + (2 + 20)


def _write_config_tree(d):
    """a config tree laid out like the reference's config/denoiser (global_config.yaml + root-level files named in its defaults
    list, `${...}` interpolation, a `trainer:` node the launch line extends with +trainer.devices / +trainer.strategy)"""
    (d / "global_config.yaml").write_text(
        "hydra:\n  run:\n    dir: .\n"
        "defaults:\n  - _self_\n  - encoder\n  - data\n  - model\n  - override hydra/hydra_logging: disabled\n"
        "project_root_path: ${hydra:runtime.cwd}\n"
        "experiment_output_path: ${project_root_path}/output/denoiser/${experiment_name}\n"
        "ckpt_path: null\nexperiment_name: null\ntrain_seed: 123\n"
        "trainer:\n  accelerator: gpu\n  max_epochs: 2000\n  check_val_every_n_epoch: ${every}\n  precision: 32\nevery: 100\n")
    (d / "encoder.yaml").write_text("ae:\n  n_embeddings: 1024\n  embedding_dim: 16\n")
    (d / "data.yaml").write_text("data:\n  batch_size: 64\n  num_workers: 10\n  data_dir: ./data/train/\n")
    (d / "model.yaml").write_text("model:\n  embed_dim: 512\n  lr_scheduler:\n    milestones: [1200, 1700]\n    gamma: 0.5\n")


def test_launch_composes_the_reference_config_layout_and_override_syntax(tmp_path):
    """scripts/train_denoiser.sh:1-7: `experiment_name=... data.batch_size=64 +trainer.devices=4 +trainer.strategy=ddp` over the
    config/denoiser tree — the subset of Hydra's composition pfpp_hip.launch implements"""
    import pytest

    for p in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from pfpp_hip import launch

    _write_config_tree(tmp_path)
    cwd = os.getcwd()
    t = launch.compose(str(tmp_path), "global_config",
                       ["experiment_name=everyday_bs64", "data.batch_size=8", "+trainer.devices=4", "+trainer.strategy=ddp",
                        "model.lr_scheduler.milestones=[3,5]"])
    assert t["trainer"] == dict(accelerator="gpu", max_epochs=2000, check_val_every_n_epoch=100, precision=32, devices=4, strategy="ddp")
    assert t["data"] == dict(batch_size=8, num_workers=10, data_dir="./data/train/") and t["ae"]["embedding_dim"] == 16
    assert t["experiment_output_path"] == f"{cwd}/output/denoiser/everyday_bs64" and t["ckpt_path"] is None
    assert t["model"]["lr_scheduler"]["milestones"] == [3, 5] and "hydra" not in t and "defaults" not in t
    with pytest.raises(KeyError):                                  # Hydra refuses a new key without the + prefix
        launch.compose(str(tmp_path), "global_config", ["trainer.devices=4"])
    # the Trainer surface train_denoiser.py:44-60 passes (**cfg.trainer): parameter sharding is refused, ddp is accepted
    tr = launch.Trainer(**t["trainer"])
    assert tr.devices == 4 and tr.strategy == "ddp" and tr.max_epochs == 2000
    with pytest.raises(ValueError):
        launch.Trainer(devices=2, strategy="fsdp")
    with pytest.raises(ValueError):
        launch.Trainer(precision="16-mixed")


def test_launch_trainer_shards_the_train_loader_like_lightning():
    """use_distributed_sampler: every rank keeps the loader's batch size / drop_last and draws from a DistributedSampler — the
    ranks' indices of an epoch are disjoint, cover the dataset, and change with set_epoch when the loader shuffled"""
    from torch.utils.data import DataLoader, DistributedSampler

    for p in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from pfpp_hip import launch

    ds = list(range(22))
    seen = []
    for rank in range(4):
        tr = launch.Trainer(devices=4, strategy="ddp", seed=5)
        tr.world_size, tr.global_rank = 4, rank
        loader, sampler = tr._shard(DataLoader(ds, batch_size=2, shuffle=True, drop_last=True), True)
        assert isinstance(sampler, DistributedSampler) and loader.batch_size == 2 and loader.drop_last
        sampler.set_epoch(0)
        e0 = [int(v) for b in loader for v in b]
        sampler.set_epoch(1)
        e1 = [int(v) for b in loader for v in b]
        assert len(e0) == 6 and e0 != e1                           # ceil(22 / 4) = 6 per rank, 3 full batches
        seen += e0
        plain, none = tr._shard(DataLoader(ds, batch_size=2, shuffle=False), False)
        padded = list(range(22)) + [0, 1]                            # DistributedSampler pads to a multiple of the world size
        assert [int(v) for b in plain for v in b] == padded[rank::4]   # unshuffled: rank, rank + 4, ...
    assert set(seen) == set(ds)
    single = launch.Trainer(devices=1)
    dl = DataLoader(ds, batch_size=2)
    assert single._shard(dl, True) == (dl, None)
