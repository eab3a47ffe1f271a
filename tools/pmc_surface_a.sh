#!/bin/bash
# L2 -> fabric read requests per k-mer of umgap_classify9_dev over the plain table and over the walk-order index: one rocprofv3 --pmc
# pass per divergence (0 and 30 per mille) over tools/bench_surface_a_walk.py --pmc (one launch of each kernel after a warm-up).
#   tools/pmc_surface_a.sh [tag]   -> gpurun_out/<tag>/m<m>/ ; summarised by tools/pmc_surface_a_summary.py
set -eo pipefail
tag=${1:-r05_surface_a}
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p "$out"
for m in 0 30; do
    rocprofv3 --pmc TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d "$out/m$m" -o pmc -- python3 tools/bench_surface_a_walk.py 50000000 10000000 $m --pmc > "$out/m$m.json" 2> "$out/m$m.err"
    echo "pass m=$m done"
done
