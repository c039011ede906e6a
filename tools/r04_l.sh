#!/bin/bash
# round 4, call l: TIMING-ONLY experiment -- fc1's result stored in a 16-row-blocked layout straight from the registers (no LDS
# transposition; -DVH_EPI_TILED_EXP=1; fc2 still reads row-major, so the logits of that library are garbage)
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=$PWD/vit-fpga_amd
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs"
for i in 1 2 3; do
  echo -n "staged (product):   "; timeout -k 10 200 python bench.py $NOX --stages 2> $out/l_stages_staged.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
  echo -n "tiled direct stores: "; VITHIP_LIB=$L/libvithip_abl_tiled.so timeout -k 10 200 python bench.py $NOX --stages 2> $out/l_stages_tiled.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
done > $out/l_tiled_exp_ab.txt 2>&1
cat $out/l_tiled_exp_ab.txt; grep fc1 $out/l_stages_staged.txt $out/l_stages_tiled.txt
