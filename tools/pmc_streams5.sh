#!/bin/bash
# Round 5: PMC passes on the VECTOR-CACHE side of the gather + write-stream microbenchmark (tools/gather_streams.py --pmc5: the pure random
# gather, + the 4-byte write stream, its LDS-staged one-writer-wave form, its 8-steps-together form): which unit holds reads back while
# writes pass?  One counter set per pass, --kernel-trace only, python started directly after `--`.  Summary: gpurun_out/<tag>/summary.json.
set -o pipefail
tag=${1:-r05_streams_pmc}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
i=0
while read -r set; do
    [ -z "$set" ] && continue
    i=$((i + 1))
    name=$(printf "p%02d" $i)
    echo "$set" > "$out/$name.set"
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/$name" -o pmc -- python3 tools/gather_streams.py --pmc5 > "$out/$name.log" 2>&1 || echo "pass $name ($set) failed" | tee -a "$out/failed.txt"
    echo "pmc pass $name done: $set"
done <<'SETS'
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TA_BUSY_sum
SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_ANY
TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_REQUEST_sum
TD_TC_STALL_sum TD_TD_BUSY_sum TD_STORE_WAVEFRONT_sum TD_LOAD_WAVEFRONT_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_CYCLE_sum TCC_EA0_WRREQ_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_LFIFO_FULL_sum
SETS
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "p??"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "k_gather_rw" not in k and "k_gather_w_variant" not in k:
                continue
            name = k[k.index("k_gather_"):].split("(")[0]
            res.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "k_gather_rw" in k or "k_gather_w_variant" in k:
                name = k[k.index("k_gather_"):].split("(")[0]
                res.setdefault(name, {}).setdefault("duration_ms_under_pmc", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
summary = {k: {c: (min(v) if c.startswith("duration") else sum(v) / len(v)) for c, v in cs.items()} for k, cs in res.items()}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary)[:1500])
PY
