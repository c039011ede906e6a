#!/bin/bash
# round 4, call f: (1) K loop of the 256x128 / 64x64-wave-tile geometry emulated in the product loop (VH_MAIN_ABL=64) against
# the loop as it is; (2) super-column width and graph replay on the default line; (3) degree-6 GELU for e4m3 results
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=$PWD/vit-fpga_amd
for v in "m0:256x256 tile as is (no epilogue)" "m64:256x128 tile emulated (half MFMAs, 6 of 8 DMA pieces)" "m0:256x256 tile as is (no epilogue)" "m64:256x128 tile emulated (half MFMAs, 6 of 8 DMA pieces)"; do
  VITHIP_LIB=$L/libvithip_diag_${v%%:*}.so timeout -k 10 200 python tools/mainloop_ablation.py --label "${v#*:}" --shape fc1,qkv 2>&1 | tee -a $out/f_halftile_kloop.txt || exit 1
done
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs"
for i in 1 2; do
  for sn in default 6 0; do
    if [ $sn = default ]; then unset VH_PP_SN; else export VH_PP_SN=$sn; fi
    echo -n "VH_PP_SN=$sn: "; timeout -k 10 200 python bench.py $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
  unset VH_PP_SN
  echo -n "--graph: "; timeout -k 10 200 python bench.py $NOX --graph 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done > $out/f_sn_graph_ab.txt 2>&1
cat $out/f_sn_graph_ab.txt
timeout -k 10 300 python -m pytest tests/test_gpu_fp8.py -x -q > $out/f_fp8_tests.log 2>&1; echo "fp8 tests rc=$?"; tail -3 $out/f_fp8_tests.log
for i in 1 2; do
  echo -n "degree 10 (abl lib): "; VITHIP_LIB=$L/libvithip_abl_deg10.so timeout -k 10 200 python bench.py $NOX --dtype fp8 --stages 2> $out/f_fp8_deg10.stages | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; grep -i "fc1" $out/f_fp8_deg10.stages | head -1
  echo -n "degree 6 (product):  "; timeout -k 10 200 python bench.py $NOX --dtype fp8 --stages 2> $out/f_fp8_deg6.stages | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; grep -i "fc1" $out/f_fp8_deg6.stages | head -1
done > $out/f_fp8_gelu_ab.txt 2>&1
cat $out/f_fp8_gelu_ab.txt
