#!/bin/bash
# tools/isa.sh <csrc file stem> [extra hipcc flags]  -- compile one kernel source to /tmp/vh_isa/<stem>*.s (never into the tree)
# and print per-kernel register / spill / scratch numbers.  Then: tools/isa.sh --body <stem> <mangled-name-regex> > k.s
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/vh_isa
mkdir -p $OUT
if [ "$1" = "--body" ]; then
    S=$OUT/$2-hip-amdgcn-amd-amdhsa-gfx950.s
    L=$(grep -n -E "^_Z.*$3.*:" $S | head -1 | cut -d: -f1)
    [ -n "$L" ] || { echo "no kernel matches $3" >&2; exit 1; }
    sed -n "${L},\$p" $S | awk '{print} /s_endpgm/{exit}'
    exit 0
fi
STEM=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result "$@" \
    -c $ROOT/vit-fpga_amd/csrc/$STEM.hip -o $OUT/$STEM.o -save-temps=obj
S=$OUT/$STEM-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "\.name:|\.vgpr_count|\.vgpr_spill_count|\.private_segment_fixed_size|\.sgpr_count" $S | paste - - - - - | \
    awk '{n=$2; sub(/^_ZN2vh/,"",n); printf "%-110s scratch %s sgpr %s vgpr %s spill %s\n", substr(n,1,110), $4, $6, $8, $10}'
