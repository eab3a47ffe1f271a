#!/bin/bash
# rocprofv3 kernel trace + stats of the sharded step at world 1 (tools/bench_sharded.py: one process, RCCL communicator of one
# rank).  UMGAP_SHARD_NO_LOCAL=1 in the environment sends every key through pack -> probe -> gather (the cost of the other
# ranks' keys at N > 1); PIECES=1 serialises the stages (stand-alone times).  Output: gpurun_out/<tag>/.
# python is started directly after `--`.
set -eo pipefail
tag=${1:-r03_sharded}
pieces=${PIECES:-8}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o stats -- python tools/bench_sharded.py --pieces "$pieces" --steps 3 > "$out/stats.log" 2>&1
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
grep '^{' "$out/stats.log" | tail -1 | cut -c1-400
