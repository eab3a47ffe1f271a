#!/usr/bin/env python3
"""Random-sector gather ceiling of the MI355X memory system as a function of table footprint and
bytes per probe (the second roofline denominator of SURVEY.md §8d), plus the probe kernel itself
(quad vs lane variant) on the same tables.  Prints one JSON line per table size."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import synth

n_tax = 100_000
out = []
for n_keys in (16_000_000, 75_000_000, 300_000_000, 1_200_000_000, 4_800_000_000):
    w = synth.SyntheticWorkload(n_tax=n_tax, n_keys=n_keys)
    n = 1 << 27
    d_out = torch.empty(n, dtype=torch.int32, device="cuda")
    row = {"table_GB": round(w.index.table_bytes / 1e9, 2), "gather_GBps": {}, "gather_Gprobes": {}}
    for bpp in (16, 32, 64, 128, 256):
        w.index.bench_gather_dev(n, bpp, d_out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            w.index.bench_gather_dev(n, bpp, d_out)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        row["gather_GBps"][bpp] = round(n * bpp / ms / 1e6, 1)
        row["gather_Gprobes"][bpp] = round(n / ms / 1e6, 2)
    keys = w.reads(0, 1_000_000)
    hits = torch.empty(keys.numel(), dtype=torch.int32, device="cuda")
    for variant in ("quad", "lane"):
        w.index.lookup9_dev(keys, hits, variant=variant)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            w.index.lookup9_dev(keys, hits, variant=variant)
        b.record()
        torch.cuda.synchronize()
        row["lookup_%s_Glookups" % variant] = round(keys.numel() / (a.elapsed_time(b) / 3) / 1e6, 2)
    print(json.dumps(row), flush=True)
    del w, d_out, keys, hits
    torch.cuda.empty_cache()
