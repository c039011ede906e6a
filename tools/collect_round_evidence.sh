#!/bin/bash
# One GPU-box call that collects the round's evidence files under gpurun_out/<tag>_* (copy what you keep into profiles/).
# Usage (repo root, on the GPU box):  bash tools/collect_round_evidence.sh r02c
set -e
tag=$1; out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
D=$PWD/vit-fpga_amd/libvithip_diag.so
timeout -k 10 400 python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
echo "default bench done"; cut -c1-220 $out/${tag}_bench_default.json
rm -rf $out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -- python3 bench.py --no-cpu-baseline --no-parity --no-fp16-line > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_bench_under_rocprof.err
cp $out/prof_$tag/*/*_kernel_stats.csv $out/${tag}_kernel_stats.csv
echo "kernel stats done"
bash tools/pmc_passes.sh $tag > $out/${tag}_pmc_summary.txt
python3 tools/pmc_traffic.py $tag $out/pmc_${tag}_fetch $out/pmc_${tag}_write > $out/${tag}_traffic.log && cp profiles/${tag}_fc1_traffic.json $out/
echo "pmc done"
timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp16-line --dtype fp8 --stages > $out/${tag}_bench_fp8.json 2> $out/${tag}_bench_fp8_stages.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --config vit_large_384 --dtype fp16 --batch 256 --steps 5 --warmup 1 --stages > $out/${tag}_bench_vitl384.json 2> $out/${tag}_bench_vitl384_stages.txt
timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --stages > /dev/null 2> $out/${tag}_bench_stages.txt
echo "config lines done"
VITHIP_LIB=$D timeout -k 10 200 python tools/gemm_anatomy.py > $out/${tag}_gemm_anatomy.txt 2>&1
VITHIP_LIB=$D timeout -k 10 100 python tools/attn_anatomy.py > $out/${tag}_attn_anatomy.txt 2>&1
VITHIP_LIB=$D timeout -k 10 100 python tools/attn_anatomy.py --config vit_large_384 --batch 256 --dtype fp16 >> $out/${tag}_attn_anatomy.txt 2>&1
timeout -k 10 100 python tools/attn_bench.py > $out/${tag}_attn_bench.txt 2>&1
timeout -k 10 100 python tools/attn_bench.py --config vit_large_384 --batch 256 --dtype fp16 >> $out/${tag}_attn_bench.txt 2>&1
echo "anatomy done"
timeout -k 10 400 python tools/parity_stats.py > $out/${tag}_parity_stats.txt 2>&1
PARITY_FOLD=off timeout -k 10 400 python tools/parity_stats.py >> $out/${tag}_parity_stats.txt 2>&1
timeout -k 10 200 python tools/torch_matmul_calib.py > $out/${tag}_torch_matmul_calib.txt 2>&1
echo "all done"
