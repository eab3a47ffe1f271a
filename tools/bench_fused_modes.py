#!/usr/bin/env python3
"""The fused 9-mer kernels per strategy on the headline batches (10 M pairs x 84 packed 9-mers, table bound to the taxonomy): LCA* -l2
(k_classify9), tree hybrid f 0.25 -l1 and MRTL -l1 behind seedextend (k_classify9_group: umgap-analyse.sh:277-288), HIP-event time per
call and a checksum of the answers -- for same-box A/B of library builds (tools/ab_walk.sh style: run it from each build's directory).
    python tools/bench_fused_modes.py [n_keys = 2000000000] [n_pairs = 10000000]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import synth
from umgap_amd.api import classify9_dev

n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
w = synth.SyntheticWorkload(n_tax=2_500_000, n_keys=n_keys, id_space=3_500_000, load_factor=0.5)
w.index.bind(w.taxonomy)
keys = [w.reads(b * B, B) for b in range(3)]
d_t = torch.empty(B, dtype=torch.int32, device="cuda")
d_s = torch.empty_like(d_t)
presets = {"lca_star_l2": dict(method="tree", strategy="lca*", lower_bound=2.0, fragments_per_read=2, min_seed_size=3, max_gap_size=1),
           "tree_hybrid_f0.25_l1": dict(method="tree", strategy="hybrid", factor=0.25, lower_bound=1.0, fragments_per_read=2, min_seed_size=3, max_gap_size=1),
           "mrtl_l1": dict(method="rmq", strategy="mrtl", lower_bound=1.0, fragments_per_read=2, min_seed_size=2, max_gap_size=1),
           "rmq_hybrid_f0.25_l1": dict(method="rmq", strategy="hybrid", factor=0.25, lower_bound=1.0, fragments_per_read=2, min_seed_size=3, max_gap_size=1)}
res = {"n_keys": n_keys, "pairs": B}
for rnd in range(2):
    for name, kw in presets.items():
        classify9_dev(w.index, w.taxonomy, keys[0], B, 84, d_t, d_s, **kw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(6):
            classify9_dev(w.index, w.taxonomy, keys[i % 3], B, 84, d_t, d_s, **kw)
        b.record()
        torch.cuda.synchronize()
        res.setdefault(name, {"ms": []})["ms"].append(round(a.elapsed_time(b) / 6, 3))
        res[name]["checksum"] = int((d_t.to(torch.int64) * 31 + d_s.to(torch.int64)).sum().item())
print(json.dumps(res))
