#!/bin/bash
# round-3 experiment 2: are the persistent workgroups' epilogues in step across the chip, and does a start-time stagger help?
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
L=$PWD/vit-fpga_amd
VITHIP_LIB=$L/libvithip_diag1.so timeout -k 10 200 python tools/gemm_anatomy.py > $O/e2_anat_t1.txt 2>&1 || exit 1
VITHIP_LIB=$L/libvithip_diag8.so timeout -k 10 200 python tools/gemm_anatomy.py > $O/e2_anat_t8.txt 2>&1 || exit 1
VH_PP_STAGGER=4,3 VITHIP_LIB=$L/libvithip_diag8.so timeout -k 10 200 python tools/gemm_anatomy.py > $O/e2_anat_t8_s43.txt 2>&1 || exit 1
grep -h "^ 1008\|epilogue start\|wave 0\|wave 4" $O/e2_anat_t1.txt $O/e2_anat_t8.txt $O/e2_anat_t8_s43.txt
for s in 0 4,3 4,6 2,6 8,2 8,4 0; do
  VH_PP_STAGGER=$s timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/e2_bench_$s.json 2> $O/e2_bench_$s.err || exit 1
  python - $s <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r03/e2_bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("stagger", sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["fp16"]["value"] if "fp16" in d else None, flush=True)
PY
done
