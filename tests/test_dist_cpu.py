"""CPU suite: the N>1 protocol of the hot path with world_size 2 on gloo.

Exactly the code path bench.py runs under torch.distributed (vh_dist: init, one broadcast of the
canonical weight blob from rank 0, contiguous image shards, max-over-ranks timing, optional gather),
with the CPU oracle standing in for the device forward (the oracle is test infrastructure; on the GPU
box the same protocol drives libvithip.so).  Checks sharded == unsharded bitwise."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, global_batch, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python"))
    import torch
    import oracle_lib as O
    import vh_dist
    import vh_synth as S

    torch.set_num_threads(1)
    r, w, _ = vh_dist.env_ranks()
    assert (r, w) == (rank, world)
    torch_, dist = vh_dist.init_process_group("gloo", rank, world)
    cfg = S.CONFIGS["vit_micro"]
    nbytes = 64 + 4 * S.param_count(cfg)
    blob = torch.zeros(nbytes, dtype=torch.uint8)
    if rank == 0:
        blob.copy_(torch.from_numpy(S.make_blob(cfg, 3)))
    vh_dist.broadcast_blob(dist, blob, src=0)
    blob_np = blob.numpy()
    assert np.array_equal(blob_np, S.make_blob(cfg, 3))          # every rank holds rank 0's weights
    images = S.make_images(cfg, 4, global_batch)                  # the global batch (synthetic, seeded)
    lo, hi = vh_dist.shard_bounds(global_batch, world, rank)
    local = O.vit_forward(cfg, blob_np, images[lo:hi], threads=2)
    t_max = vh_dist.max_over_ranks(torch_, dist, 1.0 + rank, "cpu")
    assert t_max == float(world)
    if global_batch % world == 0:
        allrows = vh_dist.gather_rows(torch_, dist, torch.from_numpy(local), world).numpy()
        if rank == 0:
            np.save(os.path.join(out_dir, "gathered.npy"), allrows)
    np.save(os.path.join(out_dir, f"local{rank}.npy"), local)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [4, 5])
def test_sharded_forward_equals_unsharded(tmp_path, global_batch):
    import torch.multiprocessing as mp

    import oracle_lib as O
    import vh_dist
    import vh_synth as S

    O.build()
    world = 2
    port = 29600 + (os.getpid() % 300) + global_batch
    mp.spawn(_worker, args=(world, port, global_batch, str(tmp_path)), nprocs=world, join=True)
    cfg = S.CONFIGS["vit_micro"]
    full = O.vit_forward(cfg, S.make_blob(cfg, 3), S.make_images(cfg, 4, global_batch), threads=2)
    parts = [np.load(tmp_path / f"local{r}.npy") for r in range(world)]
    assert sum(p.shape[0] for p in parts) == global_batch
    assert np.array_equal(np.concatenate(parts), full)           # sharded == unsharded, bitwise
    if global_batch % world == 0:
        assert np.array_equal(np.load(tmp_path / "gathered.npy"), full)


def test_shard_bounds_cover_the_batch_exactly():
    sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python"))
    import vh_dist
    for gb in (1, 7, 8, 512, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [vh_dist.shard_bounds(gb, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert vh_dist.shard_bounds(4096, 8, 3) == (1536, 2048)   # BASELINE config 3: 512 images per GPU
