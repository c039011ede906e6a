#!/bin/bash
# GPU-box calls that collect a round's evidence files under gpurun_out/<tag>_* (copy what you keep into profiles/).
# Usage (repo root, on the GPU box; three calls, each well inside gpurun's 20-minute limit):
#     bash tools/collect_round_evidence.sh r03e 1     # default bench, rocprofv3 kernel stats, PMC passes, fc1 traffic + clock files
#     bash tools/collect_round_evidence.sh r03e 2     # config lines, anatomy, attention, calibration, parity statistics
#     bash tools/collect_round_evidence.sh r03e 3     # host path, torchrun + nccl with one rank, the final default line, soak
# Parts 2 and 3 read the traffic / clock files part 1 wrote into profiles/ (they carry the hash of the kernel sources).
set -e
tag=$1; part=${2:-1}; out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
D=$PWD/vit-fpga_amd/libvithip_diag.so
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs"
# the stamped diagnostic library must be of THIS tree (a stale one misses new symbols and measures old kernels)
if [ "$part" != 3 ]; then make -s -C vit-fpga_amd diag -j8 > /dev/null 2>&1 || { echo "make diag failed"; exit 1; }; fi
if [ "$part" = 1 ]; then
timeout -k 10 400 python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
echo "default bench done"; cut -c1-220 $out/${tag}_bench_default.json
rm -rf $out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -- python3 bench.py $NOX > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_bench_under_rocprof.err
cp $out/prof_$tag/*/*_kernel_stats.csv $out/${tag}_kernel_stats.csv
echo "kernel stats done"
bash tools/pmc_passes.sh $tag > $out/${tag}_pmc_summary.txt
python3 tools/pmc_traffic.py $tag $out/pmc_${tag}_fetch $out/pmc_${tag}_write > $out/${tag}_traffic.log && cp profiles/${tag}_fc1_traffic.json $out/
echo "pmc done"
VITHIP_LIB=$D timeout -k 10 200 python tools/gemm_anatomy.py --clock-json profiles/${tag}_fc1_clock.json > $out/${tag}_gemm_anatomy.txt 2>&1
cp profiles/${tag}_fc1_clock.json $out/
echo "anatomy + clock done"
fi
if [ "$part" = 2 ]; then
timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp16-line --no-extra-configs --dtype fp8 --stages > $out/${tag}_bench_fp8.json 2> $out/${tag}_bench_fp8_stages.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs --config vit_large_384 --dtype fp16 --batch 256 --steps 5 --warmup 1 --stages > $out/${tag}_bench_vitl384.json 2> $out/${tag}_bench_vitl384_stages.txt
timeout -k 10 200 python bench.py $NOX --stages > /dev/null 2> $out/${tag}_bench_stages.txt
echo "config lines done"
VITHIP_LIB=$D timeout -k 10 200 python tools/epi_intrinsic.py > $out/${tag}_epilogue_intrinsic.txt 2>&1
VITHIP_LIB=$D timeout -k 10 100 python tools/attn_anatomy.py > $out/${tag}_attn_anatomy.txt 2>&1
VITHIP_LIB=$D timeout -k 10 100 python tools/attn_anatomy.py --config vit_large_384 --batch 256 --dtype fp16 >> $out/${tag}_attn_anatomy.txt 2>&1
timeout -k 10 100 python tools/attn_bench.py > $out/${tag}_attn_bench.txt 2>&1
timeout -k 10 100 python tools/attn_bench.py --config vit_large_384 --batch 256 --dtype fp16 >> $out/${tag}_attn_bench.txt 2>&1
echo "anatomy done"
timeout -k 10 400 python tools/parity_stats.py > $out/${tag}_parity_stats.txt 2>&1
PARITY_FOLD=off timeout -k 10 400 python tools/parity_stats.py >> $out/${tag}_parity_stats.txt 2>&1
timeout -k 10 200 python tools/torch_matmul_calib.py > $out/${tag}_torch_matmul_calib.txt 2>&1
echo "part 2 done"
fi
if [ "$part" = 3 ]; then
# the PCIe-inclusive rate (netFPGA.cpp:262-284 window) and the driver's launch line with one rank (torchrun + nccl)
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp16-line --no-extra-configs --host-path > $out/${tag}_bench_host_path.json 2> $out/${tag}_bench_host_path.err
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 \
    bench.py --gpus 1 --force-dist --no-cpu-baseline --no-extra-configs > $out/${tag}_bench_torchrun_nccl_1rank.json 2> $out/${tag}_bench_torchrun_nccl_1rank.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs > $out/${tag}_bench_plain_same_box.json 2> /dev/null
# the default line once more, now with roofline.traffic / attainable from the files of part 1
timeout -k 10 400 python bench.py > $out/${tag}_bench_default_final.json 2> $out/${tag}_bench_default_final.err
timeout -k 10 500 python tools/soak.py > $out/${tag}_soak.log 2>&1 || true
echo "part 3 done"
fi
