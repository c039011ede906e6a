#!/bin/bash
# round-3 experiment 4: trailing-MFMA barriers (TRAIL) and epilogue stores in flight across the tile boundary (FLY), A/B
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
L=$PWD/vit-fpga_amd
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -x -q > $O/e4_ops_tests.log 2>&1; echo "ops+fp8 tests rc=$?"; tail -3 $O/e4_ops_tests.log
for r in 1 2; do
for v in _abl_t0s0 _abl_t4s0 _abl_t0s1 ""; do
  echo "## libvithip$v.so" | tee -a $O/e4_ab.txt
  VITHIP_LIB=$L/libvithip$v.so timeout -k 10 200 python tools/torch_matmul_calib.py --no-torch --rounds 2 2>&1 | grep -v "^#" | tee -a $O/e4_ab.txt || exit 1
done
done
for v in _abl_t0s0 "" _abl_t0s0 ""; do
  VITHIP_LIB=$L/libvithip$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/e4_bench$v.json 2> $O/e4_bench$v.err || { tail -5 $O/e4_bench$v.err; exit 1; }
  python - "$v" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r03/e4_bench{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("lib", sys.argv[1] or "product", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("step_ms"), d["fp16"]["value"] if "fp16" in d else None, d["parity"]["worst"], flush=True)
PY
done
VITHIP_LIB=$L/libvithip_diag.so timeout -k 10 300 python tools/epi_intrinsic.py > $O/e4_epi_intrinsic.txt 2>&1; cat $O/e4_epi_intrinsic.txt
