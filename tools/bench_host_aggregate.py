#!/usr/bin/env python3
"""The host-fed taxa2agg: umgap_aggregate (taxa of every read in HOST memory, CSR -> one taxon per read in host memory; what a
Rust host binds in place of the record loop of src/commands/taxa2agg.rs:160-180), the link included.  One JSON line.
    python tools/bench_host_aggregate.py [n_reads] [hits_per_read]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import _lib, synth
from umgap_amd.api import METHODS, STRATEGIES, check

L = _lib.load()


def run(n_reads=10_000_000, hits=84, steps=3):
    w = synth.SyntheticWorkload(n_tax=2_500_000, n_keys=2_000_000_000, build_index=True)
    d_keys = w.reads(0, n_reads, kmers_per_pair=hits)
    d_taxa = w.index.lookup9_dev(d_keys, miss_value=0)
    n = d_taxa.numel()
    res = {"reads": n_reads, "hits_per_read": hits, "bytes_in_per_read": hits * 4 + 8}
    for strategy, method, lb in (("lca*", "tree", 2.0), ("mrtl", "rmq", 1.0)):
        for name, pinned in (("pinned", True), ("pageable", False)):
            h_t = torch.empty(n, dtype=torch.int32, pin_memory=pinned)
            h_t.copy_(d_taxa.cpu())
            h_o = torch.arange(0, n + 1, hits, dtype=torch.int64)
            o_t, o_s = torch.empty(n_reads, dtype=torch.int32), torch.empty(n_reads, dtype=torch.int32)
            if pinned:
                h_o, o_t, o_s = h_o.pin_memory(), o_t.pin_memory(), o_s.pin_memory()

            def call():
                check(L.umgap_aggregate(w.taxonomy._h, METHODS[method], STRATEGIES[strategy], 0.25, lb, 0, h_t.data_ptr(), None, h_o.data_ptr(), n_reads, o_t.data_ptr(), o_s.data_ptr()))
            call()
            t0 = time.perf_counter()
            for _ in range(steps):
                call()
            dt = (time.perf_counter() - t0) / steps
            res["%s %s" % (strategy, name)] = {"s_per_call": round(dt, 4), "M_reads_per_s": round(n_reads / dt / 1e6, 1), "GB_per_s_in": round(n_reads * res["bytes_in_per_read"] / dt / 1e9, 2),
                                                "status_ok_frac": round(float((o_s == 0).float().mean()), 3)}
    return res


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*a)))
