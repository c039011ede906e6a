#!/bin/bash
# round 4, call k: priority flipped per 16-row block (VH_EPI_PRIO=2) against per pass (product); the pipelined head kernel
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=$PWD/vit-fpga_amd
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vit.py -x -q -k "head or logits_match_oracle or golden" > $out/k_tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/k_tests.log
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs"
for i in 1 2 3; do
  echo -n "flip per block: "; VITHIP_LIB=$L/libvithip_abl_prio2.so timeout -k 10 200 python bench.py $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  echo -n "flip per pass:  "; timeout -k 10 200 python bench.py $NOX --stages 2> $out/k_stages.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done > $out/k_prio2_ab.txt 2>&1
cat $out/k_prio2_ab.txt; grep -E "head|final" $out/k_stages.txt
