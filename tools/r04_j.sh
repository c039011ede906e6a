#!/bin/bash
# round 4, call j: finalize_stats with its loads in flight together (A/B against the sequential loop), statistics / fold tests
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=$PWD/vit-fpga_amd
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vit.py -x -q -k "stats or fold or lnfold or resid or same_bits or deterministic" > $out/j_tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/j_tests.log
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs"
for i in 1 2 3; do
  echo -n "sequential loads: "; VITHIP_LIB=$L/libvithip_abl_oldstats.so timeout -k 10 200 python bench.py $NOX --stages 2> $out/j_stages_old.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; grep ln_stats $out/j_stages_old.txt
  echo -n "batched loads:    "; timeout -k 10 200 python bench.py $NOX --stages 2> $out/j_stages_new.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; grep ln_stats $out/j_stages_new.txt
done > $out/j_finalize_stats_ab.txt 2>&1
cat $out/j_finalize_stats_ab.txt
