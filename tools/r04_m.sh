#!/bin/bash
# round 4, call m: the tiled hidden activation (fc1 -> fc2): bitwise test, A/B of the forward (VH_H_TILED=0|1, one library)
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_vit.py -x -q -k "tiled_hidden" > $out/m_tests.log 2>&1; echo "tests rc=$?"; tail -5 $out/m_tests.log
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs"
for i in 1 2 3; do for t in 0 1; do
  echo -n "VH_H_TILED=$t: "; VH_H_TILED=$t timeout -k 10 200 python bench.py $NOX --stages 2> $out/m_stages_tiled$t.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
done; done > $out/m_h_tiled_ab.txt 2>&1
cat $out/m_h_tiled_ab.txt; grep -E "fc1|fc2" $out/m_stages_tiled0.txt $out/m_stages_tiled1.txt
