#!/bin/bash
# round 4, call g: the clamp-modifier GELU (degree 9 / 8 / 6 by result type) against round 1's (two v_med3 + degree 10): operator and
# end-to-end parity tests, then bf16 / fp16 / fp8 lines of both libraries interleaved, then 64-image parity statistics
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=$PWD/vit-fpga_amd
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -x -q -k "gemm or gelu or fp8" > $out/g_ops_tests.log 2>&1; echo "ops tests rc=$?"; tail -3 $out/g_ops_tests.log
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs"
for i in 1 2; do for dt in bf16 fp16 fp8; do
  echo -n "$dt old GELU: "; VITHIP_LIB=$L/libvithip_abl_gelu10.so timeout -k 10 200 python bench.py $NOX --dtype $dt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'] if d.get('roofline') else '')"
  echo -n "$dt new GELU: "; timeout -k 10 200 python bench.py $NOX --dtype $dt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'] if d.get('roofline') else '')"
done; done > $out/g_gelu_ab.txt 2>&1
cat $out/g_gelu_ab.txt
PARITY_DTYPES=fp16,bf16 timeout -k 10 500 python tools/parity_stats.py > $out/g_parity_stats.txt 2>&1
PARITY_DTYPES=fp16,bf16 VITHIP_LIB=$L/libvithip_abl_gelu10.so timeout -k 10 500 python tools/parity_stats.py >> $out/g_parity_stats.txt 2>&1
cat $out/g_parity_stats.txt
