#!/bin/bash
# round 4, call o: W operand of every GEMM from a tiled copy (VH_W_TILED=1) against row-major W (h tiled in both): bits, then A/B
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
python - <<'PY' > $out/o_bits.txt 2>&1
import os, sys
sys.path.insert(0, "vit-fpga_amd/python"); sys.path.insert(0, "tests")
import numpy as np, vh_synth as S, vithip
cfg = S.CONFIGS["vit_base"]; B = 256
outs = []
for wt in ("0", "1"):
    os.environ["VH_W_TILED"] = wt
    for dt in (vithip.DTYPE_BF16, vithip.DTYPE_FP16):
        ctx = vithip.VitContext(cfg, dtype=dt, max_batch=B); ctx.init_weights_seeded(0)
        px = cfg["image_size"] ** 2 * cfg["channels"]
        din, dout = vithip.DeviceBuffer(B * px * 4), vithip.DeviceBuffer(B * cfg["classes"] * 4)
        ctx.fill_input_seeded(1, B, din.ptr); ctx.forward_device(din.ptr, B, dout.ptr)
        outs.append(dout.to_numpy(np.float32, (B, cfg["classes"]))); ctx.close()
print("bf16 identical:", np.array_equal(outs[0], outs[2]), "fp16 identical:", np.array_equal(outs[1], outs[3]), "finite:", np.isfinite(outs[2]).all())
PY
cat $out/o_bits.txt
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
for i in 1 2 3 4 5 6; do for t in 0 1; do
  echo -n "VH_W_TILED=$t: "; VH_W_TILED=$t timeout -k 10 200 python bench.py $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $out/o_w_tiled_ab.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/o_w_tiled_ab.txt'):
    m=re.match(r'VH_W_TILED=(\d): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in sorted(v): print('VH_W_TILED='+k, 'images/s median', st.median(x[0] for x in v[k]), 'mean', round(st.mean(x[0] for x in v[k]),1), 'fc1 ms median', st.median(x[1] for x in v[k]), 'n', len(v[k]))
PY
