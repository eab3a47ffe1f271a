#!/usr/bin/env python3
"""What limits the random-sector rate?  (VERDICT r1 #5.)  Builds the benchmark's 8 G-key table (128 GB) and runs
   gather   : the pure random 64-byte quad gather, 8 in flight per lane (k_gather_mlp<4,8>), 840 M probes per launch
   policy   : the same with every sc0/sc1/nt cache-policy combination (k_gather_policy<0..7>)
   classify : the fused headline kernel on one 10 M-pair batch
Run plain it prints the rates; run under `rocprofv3 --pmc ...` (tools/pmc_ceiling.sh) the same launches carry counters.
   footprint: the gather confined to 1/8 ... 1/16384 of the table
    python tools/ceiling_probe.py [gather|policy|footprint|classify|all] [n_keys]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from umgap_amd import synth
from umgap_amd.api import classify9_dev

what = sys.argv[1] if len(sys.argv) > 1 else "all"
n_keys = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000_000
n_tax = 2_500_000 if what in ("classify", "all") else 100_000
w = synth.SyntheticWorkload(n_tax=n_tax, n_keys=n_keys, id_space=int(n_tax * 1.4))
n = 840_000_000
res = {"table_GB": round(w.index.table_bytes / 1e9, 1), "probes_per_launch": n}


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


d_out = torch.empty(n, dtype=torch.int32, device="cuda:0")
if what in ("gather", "all"):
    ms = timed(lambda: w.index.bench_gather_dev(n, 64 | (8 << 16), d_out))
    res["gather64_inflight8_Greq_per_s"] = round(n / ms / 1e6, 2)
    ms = timed(lambda: w.index.bench_gather_dev(n, 128 | (4 << 16), d_out))
    res["gather128_inflight4_Greq_per_s"] = round(n / ms / 1e6, 2)
if what in ("footprint", "all"):
    # the same gather confined to the first table_bytes >> shift bytes: where does the rate leave the 50 G/s plateau?
    res["footprint_Greq_per_s"] = {}
    for shift in (0, 3, 6, 8, 10, 12, 14):
        ms = timed(lambda: w.index.bench_gather_dev(n, 64 | (8 << 16) | (shift << 24), d_out))
        res["footprint_Greq_per_s"]["%.4g MB" % (w.index.table_bytes / 2 ** shift / 1e6)] = round(n / ms / 1e6, 2)
if what in ("policy", "all"):
    names = ["none", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"]
    res["policy_Greq_per_s"] = {}
    for pol, name in enumerate(names):
        ms = timed(lambda: w.index.bench_gather_dev(n, 64 | (8 << 16) | ((pol + 1) << 28), d_out))
        res["policy_Greq_per_s"][name] = round(n / ms / 1e6, 2)
if what in ("classify", "all"):
    del d_out
    B = 10_000_000
    keys = w.reads(0, B)
    d_t = torch.empty(B, dtype=torch.int32, device="cuda:0")
    d_s = torch.empty(B, dtype=torch.int32, device="cuda:0")
    w.index.bind(w.taxonomy)
    ms = timed(lambda: classify9_dev(w.index, w.taxonomy, keys, B, 84, d_t, d_s, method="tree", lower_bound=2.0))
    res["classify9_bound_ms"] = round(ms, 3)
    res["classify9_lookups_per_s_G"] = round(B * 84 / ms / 1e6, 2)
print(json.dumps(res))
