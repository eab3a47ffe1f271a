#!/bin/bash
# PMC passes that name the unit limiting the random-sector rate: one counter set per pass, --kernel-trace
# only, python started directly after `--`.  Output: gpurun_out/<tag>/<set>/..., summarised by tools/pmc_ceiling_summary.py.
set -eo pipefail
tag=${1:-r02_ceiling}
what=${2:-gather}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
i=0
while read -r set; do
    [ -z "$set" ] && continue
    i=$((i + 1))
    name=$(printf "p%02d" $i)
    echo "$set" > "$out/$name.set"
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/$name" -o pmc -- python tools/ceiling_probe.py $what > "$out/$name.log" 2>&1 || echo "pass $name ($set) failed" | tee -a "$out/failed.txt"
    echo "pmc pass $name done: $set"
done <<'SETS'
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
TCC_REQ_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_DRAM_sum
TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_LATENCY_FIFO_FULL_sum
TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum TCC_HIT_sum TCC_MISS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCC_EA0_RDREQ
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE
SETS
python tools/pmc_ceiling_summary.py "$out" "$tag"
