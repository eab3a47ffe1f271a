#!/usr/bin/env python3
"""The fused kernel on the headline batches with different -l lower bounds: -l 1 keeps the unrelated hits, so almost
every read needs the range-minimum LCA query (the aggregate phase of a realistic, noisier sample)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from umgap_amd import synth
from umgap_amd.api import classify9_dev
w = synth.SyntheticWorkload(n_tax=2_500_000, n_keys=8_000_000_000, id_space=3_500_000, load_factor=0.5)
B = 10_000_000
keys = [w.reads(b * B, B) for b in range(3)]
d_t = torch.empty(B, dtype=torch.int32, device="cuda"); d_s = torch.empty_like(d_t)
res = {}
for bound in (False, True):
  if bound:
    w.index.bind(w.taxonomy)
  for lb in (2.0, 1.0):
    for frags in (0, 2):
      for _ in (0,):
        classify9_dev(w.index, w.taxonomy, keys[0], B, 84, d_t, d_s, method="tree", lower_bound=lb, fragments_per_read=frags, min_seed_size=3, max_gap_size=1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(6):
            classify9_dev(w.index, w.taxonomy, keys[i % 3], B, 84, d_t, d_s, method="tree", lower_bound=lb, fragments_per_read=frags, min_seed_size=3, max_gap_size=1)
        b.record(); torch.cuda.synchronize()
        res["%sl%g_%s_ms" % ("bound_" if bound else "", lb, "seedextend" if frags else "plain")] = round(a.elapsed_time(b) / 6, 3)
print(json.dumps(res))
