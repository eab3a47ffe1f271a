#!/bin/bash
# Profiling recipe of a round (run on the GPU box, from the repo root):
#   tools/profile_passes.sh <tag>        e.g.  tools/profile_passes.sh r01_v6
# 1. rocprofv3 --kernel-trace --stats of the default bench command        -> gpurun_out/<tag>/stats
# 2. one rocprofv3 --pmc pass per counter set (with --kernel-trace only)  -> gpurun_out/<tag>/<set>
# tools/pmc_summary.py then writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_classify.json and
# profiles/traffic.json.  python is started directly after `--` (no env / bash -c hop).
set -eo pipefail
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$out/stats.log" 2>&1
echo "stats pass done"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    name=$(echo "$set" | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/$name" -o pmc -- python bench.py --steps 2 --warmup 1 --batches 2 --no-cpu-baseline --no-gather --no-extras > "$out/$name.log" 2>&1
    echo "pmc pass $name done"
done
python tools/pmc_summary.py "$out" "$tag"
