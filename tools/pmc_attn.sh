#!/bin/bash
# SQ counters of the attention kernel alone (tools/attn_bench.py), one rocprofv3 pass per group.  Usage: bash tools/pmc_attn.sh <tag>
tag=$1; cd "$(dirname "$0")/.."; export TMPDIR=/tmp; out=gpurun_out
run() { local name=$1; shift; rm -rf $out/pmca_${tag}_$name
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmca_${tag}_$name -- python3 tools/attn_bench.py --iters 5 > $out/pmca_${tag}_$name.log 2>&1 || tail -3 $out/pmca_${tag}_$name.log; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run b SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run c SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32
run d SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS
run e GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$out/pmca_${tag}_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention" in r["Kernel_Name"]:
            t = tot[r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1
for k in sorted(tot): print(f"{k:32s} {tot[k][0] / tot[k][1]:16.0f}  ({tot[k][1]} launches)")
PY
