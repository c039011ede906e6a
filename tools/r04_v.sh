#!/bin/bash
# round 4, call v: cache policy of the split-residual epilogue's plane accesses (VH_RS_NT: 1 nt loads, 2 nt hi stores, 4 nt lo stores), interleaved
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=vit-fpga_amd
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
LIBS="libvithip.so libvithip_abl_nt1.so libvithip_abl_nt2.so libvithip_abl_nt4.so libvithip_abl_nt6.so libvithip_abl_nt7.so"
for lib in $LIBS; do echo -n "$lib: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 120 python tools/soak.py --steps 4 --every 2 2>&1 | tail -1; done | tee $out/v_hashes.txt
for i in 1 2 3 4; do for lib in $LIBS; do
  echo -n "$lib: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $out/v_nt.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/v_nt.txt'):
    m=re.match(r'(\S+): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in v: print(k, 'images/s median', st.median(x[0] for x in v[k]), 'fc1 ms median', st.median(x[1] for x in v[k]), 'n', len(v[k]))
PY
for lib in $LIBS; do VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/v_stages_$lib.txt > /dev/null; echo "$lib: $(grep -E 'qkv_gemm|proj_gemm|fc1_gemm|fc2_gemm' $out/v_stages_$lib.txt | awk '{printf "%s %s  ", $1, $2}')"; done
