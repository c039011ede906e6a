"""The driver's multi-GPU launch line on the hardware that is available: `python -m torch.distributed.run --nproc-per-node 1`
with the nccl (= RCCL) backend.  One rank is all a one-GPU box can give, but it is the real thing end to end: torchrun's
rendezvous, an RCCL communicator, the weight blob through dist.broadcast on device memory, libvithip bound under the HIP
runtime torch bundles.  The rank's logits must equal a plain context's, bit for bit.  N > 1 ranks on distinct GPUs is the
driver's run (SCALE_rNN.json); the N > 1 protocol itself is covered on CPU (tests/test_dist_cpu.py, world 2, gloo)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import vh_synth as S

pytestmark = pytest.mark.gpu
vithip = pytest.importorskip("vithip")

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("dtype_name", ["bf16", "fp16"])
def test_single_rank_torchrun_nccl_path_equals_a_plain_context_bitwise(tmp_path, dtype_name):
    cfg_name, n = "vit_tiny", 5
    port = 29700 + os.getpid() % 200
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dist_rank_worker.py"), str(tmp_path), cfg_name, str(n), dtype_name]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(tmp_path / "logits0.npy")
    cfg = S.CONFIGS[cfg_name]
    ctx = vithip.VitContext(cfg, dtype={"bf16": vithip.DTYPE_BF16, "fp16": vithip.DTYPE_FP16}[dtype_name], max_batch=n)
    ctx.init_weights_seeded(0)
    want = ctx.forward(S.make_images(cfg, 1, n))
    ctx.close()
    assert got.shape == want.shape and np.array_equal(got, want)
