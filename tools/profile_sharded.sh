#!/bin/bash
# rocprofv3 kernel trace + stats of the sharded step at world 1 (bench.py --force-dist: RCCL process group of one rank, the
# rank's own share of the exchange stays in place).  Output: gpurun_out/<tag>/, stats copied to profiles/<tag>_kernel_stats.csv
# by tools/pmc_summary-style post-processing on the build machine.  python is started directly after `--`.
set -eo pipefail
tag=${1:-r02_sharded}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o stats -- python bench.py --gpus 1 --force-dist --no-extras --no-cpu-baseline --no-gather --steps 4 --warmup 1 > "$out/stats.log" 2>&1
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
tail -1 "$out/stats.log" | cut -c1-200
