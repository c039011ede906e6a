#!/bin/bash
# round 4, call c: VALU issue-rate probe; guard / group / ViT-L tests; fp8 attention A/B (staged-Q ring for e4m3 results);
# clock vs super-column width; anatomy of a tile deep in the steady state (VH_DIAG_REC_IT=8)
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
timeout -k 10 120 tools/probe_valu > $out/c_probe_valu.txt 2>&1; cat $out/c_probe_valu.txt
timeout -k 10 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_fp8.py tests/test_gpu_group.py tests/test_gpu_ops.py -x -q -s -k "backstop or tripped_guard or massive or vit_large_384_fp16_is_inside or rehearsal or attention or fp8_emulation or e4m3" > $out/c_tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/c_tests.log
timeout -k 10 300 python bench.py --group --gpus 8 --same-device --batch 64 --steps 5 --warmup 1 --no-parity > $out/c_bench_group8_rehearsal.json 2> $out/c_bench_group8_rehearsal.err; cut -c1-300 $out/c_bench_group8_rehearsal.json
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs"
for r in 2 1 2 1; do VH_ATTN_RING=$r timeout -k 10 200 python bench.py $NOX --dtype fp8 --stages 2> $out/c_fp8_ring$r.stages | cut -c1-120; grep -i "attn\|attention" $out/c_fp8_ring$r.stages | head -3; done
make -s -C vit-fpga_amd diag -j8 > /dev/null 2>&1 || { echo "make diag failed"; exit 1; }
D=$PWD/vit-fpga_amd/libvithip_diag.so
for sn in 0 2 3 4 6 12; do echo "== VH_PP_SN=$sn"; VH_PP_SN=$sn VITHIP_LIB=$D timeout -k 10 100 python tools/gemm_anatomy.py --seconds 2 2>&1 | grep -E "^ 100864 +(3072|2304) " | cut -c1-230; done > $out/c_sn_clock.txt 2>&1
cat $out/c_sn_clock.txt
touch vit-fpga_amd/csrc/kernels_gemm5.hip; make -s -C vit-fpga_amd diag -j8 DIAGFLAGS=-DVH_DIAG_REC_IT=8 > /dev/null 2>&1 || { echo "make diag (REC_IT=8) failed"; exit 1; }
VITHIP_LIB=$D timeout -k 10 100 python tools/gemm_anatomy.py --seconds 2 > $out/c_gemm_anatomy_tile8.txt 2>&1
grep -E "^ 100864|wave 0:|wave 4:" $out/c_gemm_anatomy_tile8.txt | cut -c1-200
