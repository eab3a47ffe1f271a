#!/usr/bin/env python3
"""Is there request capacity beside the vector cache?  The random 64-byte quad gather (128 vector probes per wave step, 8 in flight per
lane: the ~50 G requests/s ceiling) with 0 / 1 / 2 / 4 scalar-path probes (s_load_dwordx16 of a random bucket) per wave step beside it.
If the time does not grow, the scalar probes came for free.  One JSON line.   python tools/gather_scalar_mix.py [n_keys] [n_probes]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import synth
from umgap_amd.api import Index

n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 840_000_000
w = synth.SyntheticWorkload(n_tax=100_000, n_keys=1000, build_index=False)
ix = Index.synthetic(n_keys, w.d_taxon_of_row, synth.MUL_A, synth.ADD_B, 0.5, 0, 1)
out = torch.empty(n, dtype=torch.int32, device="cuda")
res = {"table_GB": round(ix.table_bytes / 1e9, 1), "vector_probes": n, "legs": {}}
for ks, code in ((0, 72), (1, 65), (2, 66), (4, 68), (0, 72)):
    best = None
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ix.bench_gather_dev(n, code | (8 << 16), out)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        best = ms if best is None or ms < best else best
    extra = n // 128 * ks
    res["legs"]["%d scalar probes per 128 vector probes%s" % (ks, " (again)" if ks == 0 and res["legs"] else "")] = {
        "ms": round(best, 3), "scalar_probes": extra, "G_requests_per_s_all": round((n + extra) / best / 1e6, 2), "G_vector_probes_per_s": round(n / best / 1e6, 2)}
print(json.dumps(res))
