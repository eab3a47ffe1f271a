#!/bin/bash
# PMC counters of the fused kernels for the other strategies (k_classify9_group = first pass, k_classify9_general = the
# one-read-per-wave pass for reads with > 64 distinct taxa) and of the headline kernel, from the `other_pipelines` legs of bench.py.
#   tools/pmc_general.sh [tag]   -> gpurun_out/<tag>/ ; prints the means per launch; tools/pmc_general_summary.py writes profiles/<tag>.json
set -eo pipefail
tag=${1:-r02_pmc_general}
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/p$i" -o pmc -- python bench.py --steps 2 --warmup 1 --batches 2 --no-cpu-baseline --no-gather --no-end-to-end > "$out/p$i.log" 2>&1
    echo "pass $i done"
done
python tools/pmc_general_summary.py "$out" "$tag"
