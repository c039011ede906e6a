"""One rank of the driver's launch line (python -m torch.distributed.run ... ), reduced to what the hot path does under it:
nccl (= RCCL) process group, rank 0 generates the weights and exports the canonical blob into a torch tensor,
dist.broadcast, every other rank loads the broadcast blob; each rank then runs its contiguous shard of the seeded batch
through libvithip (bound AFTER torch, i.e. under torch's bundled HIP runtime: the combination bench.py runs in) and writes
its logits.  Driven by tests/test_gpu_dist.py; not a test module itself."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "vit-fpga_amd", "python"))


def main():
    out_dir, config, global_batch, dtype_name = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    import vh_dist
    rank, world, local_rank = vh_dist.env_ranks()
    torch, dist = vh_dist.init_process_group("nccl", rank, world, local_rank)
    import numpy as np
    import vh_synth as S
    import vithip
    cfg = S.CONFIGS[config]
    dt = {"bf16": vithip.DTYPE_BF16, "fp16": vithip.DTYPE_FP16}[dtype_name]
    lo, hi = vh_dist.shard_bounds(global_batch, world, rank)
    ctx = vithip.VitContext(cfg, dtype=dt, max_batch=max(1, hi - lo), device=local_rank)
    wbuf = torch.empty(ctx.blob_bytes, dtype=torch.uint8, device=f"cuda:{local_rank}")
    if rank == 0:
        ctx.init_weights_seeded(0)
        ctx.export_weights_device(wbuf.data_ptr(), ctx.blob_bytes)
    torch.cuda.synchronize()
    vh_dist.broadcast_blob(dist, wbuf, src=0)
    torch.cuda.synchronize()
    # every rank (rank 0 too: the blob has been through RCCL) loads what the broadcast delivered
    ctx.load_weights_device(wbuf.data_ptr(), ctx.blob_bytes)
    images = S.make_images(cfg, 1, global_batch)[lo:hi]
    logits = ctx.forward(images)
    t = vh_dist.max_over_ranks(torch, dist, 1.0 + rank, f"cuda:{local_rank}")
    assert t == float(world)
    np.save(os.path.join(out_dir, f"logits{rank}.npy"), logits)
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
