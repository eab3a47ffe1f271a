#!/bin/bash
# PMC counters of the fused tryptic pipeline (tools/bench_tryptic_fused.py: umgap_classify_tryp_dev = digest -> window -> lookup ->
# CSR group aggregate): one rocprofv3 --pmc pass per counter set, with --kernel-trace only.
#   tools/pmc_tryptic.sh [tag] [n_keys] [n_pairs]   -> gpurun_out/<tag>/ ; summarise with tools/pmc_tryptic_summary.py
set -eo pipefail
tag=${1:-r02_tryptic_fused}
nk=${2:-400000000}
np=${3:-10000000}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
i=0
while read -r set; do
    [ -z "$set" ] && continue
    i=$((i + 1))
    name=$(printf "p%02d" $i)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/$name" -o pmc -- python tools/bench_tryptic_fused.py $nk $np > "$out/$name.log" 2>&1 || echo "pass $name ($set) failed" | tee -a "$out/failed.txt"
    echo "pmc pass $name done: $set"
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
SETS
