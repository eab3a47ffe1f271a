#!/usr/bin/env python3
"""The host-fed get(): umgap_index_lookup9 (packed 9-mers in HOST memory -> taxa in host memory; what a Rust host binds in place
of fst::Map::get, src/commands/pept2lca.rs:93) with the link included, from pinned and from pageable arrays.  One JSON line.
    python tools/bench_host_lookup.py [n_keys_in_index] [n_queries]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from umgap_amd import _lib, synth
from umgap_amd.api import Index, check

L = _lib.load()


def run(n_keys=2_000_000_000, n_q=200_000_000, steps=3):
    w = synth.SyntheticWorkload(n_tax=2_500_000, n_keys=n_keys, build_index=True)
    ix = w.index
    d_q = w.reads(0, n_q // synth.KMERS_PER_PAIR)          # device keys of the benchmark's mix (hits and misses)
    n = d_q.numel()
    res = {"index_keys": n_keys, "queries": n, "bytes_per_query": {"in": 8, "out": 4}}
    want = ix.lookup9_dev(d_q, miss_value=0).cpu()
    for name, pinned in (("pinned", True), ("pageable", False)):
        h_q = torch.empty(n, dtype=torch.int64, pin_memory=pinned)
        h_q.copy_(d_q.cpu())
        h_o = torch.empty(n, dtype=torch.int32, pin_memory=pinned)

        def call():
            check(L.umgap_index_lookup9(ix._h, h_q.data_ptr(), n, 0, h_o.data_ptr()))
        call()
        assert torch.equal(h_o, want)
        times = []
        for _ in range(max(steps, 5)):       # every call timed on its own: min / median / max
            t0 = time.perf_counter()
            call()
            times.append(time.perf_counter() - t0)
        times.sort()
        dt = times[len(times) // 2]
        res[name] = {"s_per_call": round(dt, 4), "G_lookups_per_s": round(n / dt / 1e9, 3), "GB_per_s_in": round(n * 8 / dt / 1e9, 2), "calls": len(times),
                     "G_lookups_per_s_min_median_max": [round(n / times[-1] / 1e9, 3), round(n / dt / 1e9, 3), round(n / times[0] / 1e9, 3)]}
    return res


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*a)))
