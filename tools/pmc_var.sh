#!/bin/bash
# Profile of the tryptic probe k_lookup_var (BASELINE configs[4]): rocprofv3 --kernel-trace --stats, then one --pmc pass per
# counter set (with --kernel-trace only), over tools/bench_config5.py.  Launches of the kernel in that script, in order: the
# build's self check (n = n_keys), 4 x the insertion-ordered queries, 4 x the shuffled 50 M-query sample.
#   tools/pmc_var.sh [tag] [n_keys] [n_queries] [uniform|tryptic]   -> gpurun_out/<tag>/ ; summarise with tools/pmc_var_summary.py on
#   the build machine.  UMGAP_VAR_INLINE=0 in the environment builds the table without inline keys (A/B).
set -eo pipefail
tag=${1:-r02_lookup_var}
nk=${2:-400000000}
nq=${3:-200000000}
lens=${4:-uniform}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o stats -- python tools/bench_config5.py $nk $nq $lens > "$out/stats.log" 2>&1
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
i=0
while read -r set; do
    [ -z "$set" ] && continue
    i=$((i + 1))
    name=$(printf "p%02d" $i)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/$name" -o pmc -- python tools/bench_config5.py $nk $nq $lens > "$out/$name.log" 2>&1 || echo "pass $name ($set) failed" | tee -a "$out/failed.txt"
    echo "pmc pass $name done: $set"
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_FLAT
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
SETS
grep -h "^{" "$out/stats.log" | tail -1
