#!/bin/bash
# PMC passes of the default bench command, one rocprofv3 run per counter group (MI355X_MICROARCH.md "rocprofv3 PMC
# slots": FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with tracing domains other than
# --kernel-trace).  Usage (on the GPU box, from the repo root):  bash tools/pmc_passes.sh <tag> [extra bench args]
# Writes gpurun_out/pmc_<tag>_{mfma,fetch,write,lds}/ and prints the summary; copy what you keep into profiles/.
set -e
tag=$1; shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out
run() {  # name, counters...
  local name=$1; shift
  rm -rf $out/pmc_${tag}_$name
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_${tag}_$name -- \
      python3 bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 3 --warmup 1 "${EXTRA[@]}" > $out/pmc_${tag}_$name.log 2>&1
}
EXTRA=("$@")
run mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL
python3 tools/pmc_summary.py $out/pmc_${tag}_mfma $out/pmc_${tag}_fetch $out/pmc_${tag}_write $out/pmc_${tag}_lds
