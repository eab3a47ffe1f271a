#!/usr/bin/env python3
"""k_lookup9 alone: 840 M packed 9-mers (the benchmark's read mix) against the 8 G-key table, hits to HBM.  One JSON line.
    python tools/bench_lookup9.py [n_keys] [n_pairs]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import synth

n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000_000
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
w = synth.SyntheticWorkload(n_tax=2_500_000, n_keys=n_keys, id_space=3_500_000, load_factor=0.5)
keys = [w.reads(b * n_pairs, n_pairs) for b in range(2)]
out = torch.empty(n_pairs * synth.KMERS_PER_PAIR, dtype=torch.int32, device="cuda")
w.index.lookup9_dev(keys[0], out)
torch.cuda.synchronize()
ms = []
for i in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    w.index.lookup9_dev(keys[i % 2], out)
    b.record()
    torch.cuda.synchronize()
    ms.append(a.elapsed_time(b))
n = n_pairs * synth.KMERS_PER_PAIR
print(json.dumps({"kernel": "k_lookup9", "lookups": n, "table_GB": round(w.index.table_bytes / 1e9, 1), "ms_min_median_max": [round(min(ms), 3), round(sorted(ms)[3], 3), round(max(ms), 3)],
                  "G_lookups_per_s": round(n / sorted(ms)[3] / 1e6, 2), "hit_fraction": round(float((out != 0).float().mean()), 3)}))
