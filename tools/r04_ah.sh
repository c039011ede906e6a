#!/bin/bash
# round 4, call ah: the prefetching epilogues with their choice made at compile time (no second path; libvithip_abl_norpre.so = this tree with
# -DVH_PP_RPRE=0) against the previous commit (libvithip_abl_old.so: run-time flag), interleaved
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=vit-fpga_amd
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
for lib in libvithip_abl_old.so libvithip_abl_norpre.so; do echo -n "$lib: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 120 python tools/soak.py --steps 40 --every 20 2>&1 | tail -1; done | tee $out/ah_hashes.txt
for i in 1 2 3 4 5 6; do for lib in libvithip_abl_old.so libvithip_abl_norpre.so; do
  echo -n "$lib: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $out/ah_cpre_compile_time.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/ah_cpre_compile_time.txt'):
    m=re.match(r'(\S+): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in sorted(v): print(k, 'images/s median', st.median(x[0] for x in v[k]), 'fc1', st.median(x[1] for x in v[k]), [x[0] for x in v[k]])
PY
