#!/bin/bash
# same-box A/B of the surface-B walk kernels across library builds kept in sub-directories (git worktrees with their own
# libumgap_hip.so):  tools/ab_walk.sh "<dir> <dir> ..." "<mutations per mille> ..." [n_fragments]
set -eo pipefail
root=$(pwd)
nf=${3:-50000000}
for m in $2; do
    for v in $1; do
        name=$(basename "$(cd "$v" && pwd)")
        (cd "$v" && timeout -k 10 300 python tools/bench_surface_b.py $nf 10000000 $m --walk) > "$root/gpurun_out/ab_${name}_m$m.json" 2> "$root/gpurun_out/ab_${name}_m$m.err"
        echo "$name m=$m done"
    done
done
