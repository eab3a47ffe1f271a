#!/usr/bin/env python3
"""Random 64-byte gather rate over the full-size table as a function of the loads a lane keeps in flight."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from umgap_amd import synth
n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000_000
w = synth.SyntheticWorkload(n_tax=100_000, n_keys=n_keys)
n = 840_000_000
d_out = torch.empty(n, dtype=torch.int32, device="cuda:0")
res = {"table_GB": round(w.index.table_bytes / 1e9, 1)}
for label, code in (("1", 64), ("2", 64 | (2 << 16)), ("4", 64 | (4 << 16)), ("8", 64 | (8 << 16))):
    w.index.bench_gather_dev(n, code, d_out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    w.index.bench_gather_dev(n, code, d_out)
    b.record()
    torch.cuda.synchronize()
    res["in_flight_%s_Greq_per_s" % label] = round(n / (a.elapsed_time(b) * 1e-3) / 1e9, 2)
print(json.dumps(res))
