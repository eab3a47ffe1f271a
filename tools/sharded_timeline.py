#!/usr/bin/env python3
"""Timeline of the last sharded step out of a rocprofv3 --kernel-trace of tools/bench_sharded.py (tools/profile_sharded.sh):
which of pack / probe / aggregate ran when, and the bench line's sharded figures.   tools/sharded_timeline.py gpurun_out/<tag> [out.txt]"""
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
trace = glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(trace)) if any(k in r["Kernel_Name"] for k in ("k_shard_pack", "k_shard_probe", "k_classify9<false, true>"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_pack = sum("k_shard_pack" in r["Kernel_Name"] for r in rows)
n_probe = sum("k_shard_probe" in r["Kernel_Name"] for r in rows)
pieces = max(1, n_probe // 4)        # bench_sharded.py: one check step + three timed steps
last = rows[-(pieces * (2 + (1 if n_pack else 0))):]
t0 = int(last[0]["Start_Timestamp"])
lines = []
for line in open(os.path.join(src, "stats.log"), errors="replace"):
    if line.startswith("{") and '"sharded_ms_per_step"' in line:
        d = json.loads(line)
        lines.append("under the profiler: sharded %.2f ms per step (%d pieces, own keys %s), whole table %.2f ms" % (d["sharded_ms_per_step"], d["pieces"], d["own_keys"], d["whole_table_ms_per_step"]))
lines.append("kernel     queue    start ->     end   (ms, last step)")
for r in last:
    n = r["Kernel_Name"]
    name = "pack" if "pack" in n else "probe" if "probe" in n else "aggregate"
    lines.append("%-10s q%s  %8.2f -> %8.2f  (%.2f)" % (name, r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6,
                                                       (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text)
