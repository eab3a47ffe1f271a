#!/bin/bash
# Profile of the surface-B kernel k_classify_prot9 (protein fragments in): rocprofv3 --kernel-trace --stats, then one --pmc
# pass per counter set (with --kernel-trace only) over tools/bench_surface_b.py (100 M-fragment proteome = 4.2 G keys, 10 M pairs).
#   tools/pmc_prot.sh [tag] [bench_surface_b arguments, e.g. "50000000 10000000 30 --walk"]   -> gpurun_out/<tag>/ ; tools/pmc_prot_summary.py writes profiles/<tag>_{stats.csv,pmc.json} on the build machine
set -eo pipefail
tag=${1:-r02_prot9}
args=${2:-}
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o stats -- python tools/bench_surface_b.py $args > "$out/stats.log" 2>&1
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/p$i" -o pmc -- python tools/bench_surface_b.py $args > "$out/p$i.log" 2>&1
    echo "pass $i done"
done
grep -h "^{" "$out/stats.log" | tail -1 | cut -c1-300
