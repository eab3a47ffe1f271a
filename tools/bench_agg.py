#!/usr/bin/env python3
"""Aggregation kernels alone (hits resident in HBM): reads/s per strategy for reads of 16 hits (tryptic
pipelines) and 84 hits (the 9-mer sensitivity pipelines).  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd.synth import SyntheticWorkload

dev = torch.device("cuda", 0)
w = SyntheticWorkload(n_tax=2_500_000, n_keys=64_000_000)
res = {}
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
for stride, n_reads in ((16, 10_000_000), (84, 4_000_000)):
    keys = w.reads(0, n_reads, kmers_per_pair=stride)
    hits = w.index.lookup9_dev(keys)
    dt = torch.empty(n_reads, dtype=torch.int32, device=dev); ds = torch.empty_like(dt)
    for name, kw in (("mrtl_l1", dict(method="rmq", strategy="mrtl", lower_bound=1.0)),
                     ("hybrid_f0.25_l1", dict(method="tree", strategy="hybrid", factor=0.25, lower_bound=1.0)),
                     ("rmq_hybrid_f0.25", dict(method="rmq", strategy="hybrid", factor=0.25, lower_bound=1.0)),
                     ("lca_star_l2", dict(method="tree", strategy="lca*", lower_bound=2.0))):
        w.taxonomy.aggregate_strided_dev(hits, stride, n_reads, dt, ds, **kw)
        torch.cuda.synchronize()
        t0.record()
        for _ in range(3):
            w.taxonomy.aggregate_strided_dev(hits, stride, n_reads, dt, ds, **kw)
        t1.record(); torch.cuda.synchronize()
        res["%d_hits_%s_Mreads_per_s" % (stride, name)] = round(n_reads / (t0.elapsed_time(t1) / 3) / 1e3, 1)
    del keys, hits
print(json.dumps(res))
