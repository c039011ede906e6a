#!/bin/bash
# round 4, call h: epilogue priority alternating between the two wave groups (VH_EPI_PRIO) against none; anatomy at tile 8
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=$PWD/vit-fpga_amd
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs"
for i in 1 2 3; do for dt in bf16 fp16; do
  echo -n "$dt no prio:  "; VITHIP_LIB=$L/libvithip_abl_noprio.so timeout -k 10 200 python bench.py $NOX --dtype $dt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'] if d.get('roofline') else '')"
  echo -n "$dt alt prio: "; timeout -k 10 200 python bench.py $NOX --dtype $dt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'] if d.get('roofline') else '')"
done; done > $out/h_prio_ab.txt 2>&1
cat $out/h_prio_ab.txt
touch vit-fpga_amd/csrc/kernels_gemm5.hip; make -s -C vit-fpga_amd diag -j8 DIAGFLAGS=-DVH_DIAG_REC_IT=8 > /dev/null 2>&1 || { echo "make diag (REC_IT=8) failed"; exit 1; }
VITHIP_LIB=$L/libvithip_diag.so timeout -k 10 100 python tools/gemm_anatomy.py --seconds 2 > $out/h_gemm_anatomy_tile8_prio.txt 2>&1
grep -E "^ 100864|wave 0:|wave 4:" $out/h_gemm_anatomy_tile8_prio.txt | cut -c1-200
