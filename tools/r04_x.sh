#!/bin/bash
# round 4, call x: q|k|v head-major between the projection's epilogue and attention's operand DMA (VH_QKV_HM): bits, then interleaved A/B
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=vit-fpga_amd
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
python -m pytest tests/test_gpu_vit.py -x -q -m gpu -k "tiled or bits" > $out/x_tests.txt 2>&1 || { tail -15 $out/x_tests.txt; exit 1; }
tail -1 $out/x_tests.txt
for hm in 0 1; do for dt in bf16 fp16; do echo -n "VH_QKV_HM=$hm $dt: "; VH_QKV_HM=$hm timeout -k 10 120 python tools/soak.py --steps 4 --every 2 --dtype $dt 2>&1 | tail -1; done; done | tee $out/x_hashes.txt
for i in 1 2 3 4 5; do for hm in 0 1; do for dt in bf16 fp16; do
  echo -n "hm$hm-$dt: "; VH_QKV_HM=$hm timeout -k 10 200 python bench.py --dtype $dt $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done; done > $out/x_qkv_hm.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/x_qkv_hm.txt'):
    m=re.match(r'(\S+): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in sorted(v): print(k, 'images/s median', st.median(x[0] for x in v[k]), 'n', len(v[k]), [x[0] for x in v[k]])
PY
for hm in 0 1; do VH_QKV_HM=$hm timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/x_stages_hm$hm.txt > /dev/null; echo "hm$hm: $(grep -E 'qkv_gemm|attention|proj_gemm' $out/x_stages_hm$hm.txt | awk '{printf "%s %s  ", $1, $2}')"; done
for hm in 0 1; do VH_QKV_HM=$hm timeout -k 10 300 python bench.py --config vit_large_384 --dtype fp16 --batch 256 --steps 5 --warmup 1 --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/x_stages_vitl_hm$hm.txt > $out/x_vitl_hm$hm.json; echo "vitl hm$hm: $(python -c "import json; print(json.load(open('$out/x_vitl_hm$hm.json'))['value'])") $(grep -E 'qkv_gemm|attention|proj_gemm' $out/x_stages_vitl_hm$hm.txt | awk '{printf "%s %s  ", $1, $2}')"; done
