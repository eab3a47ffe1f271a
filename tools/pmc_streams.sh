#!/bin/bash
# PMC passes over the gather + stream microbenchmark (tools/gather_streams.py --pmc: the pure random gather, + a 16-byte read stream,
# + a 4-byte and a 16-byte write stream per probe): which unit does a write stream hold up?  One counter set per pass,
# --kernel-trace only, python started directly after `--`.  Output: gpurun_out/<tag>/pNN/, summarised into gpurun_out/<tag>/summary.json.
set -o pipefail
tag=${1:-r04_streams_pmc}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
i=0
while read -r set; do
    [ -z "$set" ] && continue
    i=$((i + 1))
    name=$(printf "p%02d" $i)
    echo "$set" > "$out/$name.set"
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/$name" -o pmc -- python tools/gather_streams.py --pmc > "$out/$name.log" 2>&1 || echo "pass $name ($set) failed" | tee -a "$out/failed.txt"
    echo "pmc pass $name done: $set"
done <<'SETS'
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum
TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum
TCC_WRITE_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_CYCLE_sum
TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum
SETS
python - "$out" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "p??"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "k_gather_rw" not in k:
                continue
            name = k[k.index("k_gather_rw"):].split("(")[0]
            res.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary)[:3000])
PY
