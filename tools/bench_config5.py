#!/usr/bin/env python3
"""BASELINE.json configs[4]: tryptic-peptide (variable-length key) index on 1 x MI355X and the
aggregation strategies the tryptic / sensitivity pipelines use (MRTL -l1/-l5, hybrid f=0.25).
Everything is generated on the device.  Prints one JSON line."""
import ctypes as C
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from umgap_amd import _lib, synth
from umgap_amd.api import Index, _dev_ptr, _stream_ptr

dev = torch.device("cuda", 0)
n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000_000
LENGTHS = sys.argv[3] if len(sys.argv) > 3 else "uniform"        # "uniform": 14..45; "tryptic": 14 + geometric, most keys short
L = _lib.load()


def splitmix(x):
    x = x + torch.tensor(-7046029254386353131, dtype=torch.int64, device=dev)  # 0x9E3779B97F4A7C15
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * torch.tensor(-4658895280553007687, dtype=torch.int64, device=dev)
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * torch.tensor(-7723592293110705685, dtype=torch.int64, device=dev)
    return x ^ ((x >> 31) & 0x1FFFFFFFF)


def peptides(first, n, seed):
    idx = torch.arange(first, first + n, dtype=torch.int64, device=dev)
    u = splitmix(idx ^ seed) & 0x7FFFFFFF
    if LENGTHS == "tryptic":
        # a cleavage site every ~10 residues: 14 + geometric(p = 0.1), at most 45 (no shorter than 14: the generator's peptides end in
        # K or R, and random shorter ones collide at 1 G keys).  34 % of the keys have at most 17 residues and live inside their slot;
        # among real tryptic peptides of 5..50 residues it is about 70 %.
        g = torch.floor(torch.log((u.to(torch.float64) + 0.5) / 2147483648.0) / math.log(0.9)).to(torch.int64)
        lens = 14 + torch.clamp(g, max=31)
    else:
        lens = 14 + u % 32                                        # 14..45 residues (12-mers would collide at 1 G keys)
    offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=offs[1:])
    total = int(offs[-1].item())
    data = torch.empty(total, dtype=torch.uint8, device=dev)
    _lib.check(L.umgap_synth_peptides_dev(first, n, seed, _dev_ptr(offs), _dev_ptr(data), 0, _stream_ptr(None)))
    return data, offs, total


w = synth.SyntheticWorkload(n_tax=2_500_000, n_keys=1000, build_index=False)
res = {"n_keys": n_keys, "n_queries": n_q, "key_lengths": LENGTHS}
data, offs, total = peptides(0, n_keys, 43)
taxa = w.d_taxon_of_row[torch.arange(n_keys, device=dev) % w.n_tax].contiguous()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
h = C.c_void_p()
torch.cuda.synchronize()
t0.record()
_lib.check(L.umgap_index_build_var_dev(_dev_ptr(data), _dev_ptr(offs), _dev_ptr(taxa), n_keys, total, 0.0, 0, C.byref(h)))
t1.record(); torch.cuda.synchronize()
ix = Index(h.value)
res["layout"] = "keys in the slots (1-3 per key, no heap)" if ix.var_layout == 1 else "slots + heap"
res["table_GB"] = round(ix.table_bytes / 1e9, 2)
res["key_bytes_GB"] = round(total / 1e9, 2)
res["build_s"] = round(t0.elapsed_time(t1) / 1e3, 2)
del data, offs
torch.cuda.empty_cache()
# queries: 70 % members (regenerated from their number), 30 % strangers (another seed)
nm = int(n_q * 0.7)
qd_m, qo_m, tm = peptides(0, nm, 43)          # members 0..nm-1 (nm <= n_keys)
qd_s, qo_s, ts = peptides(10**12, n_q - nm, 99)
qd = torch.cat([qd_m, qd_s]); qo = torch.cat([qo_m, qo_s[1:] + tm])
del qd_m, qd_s, qo_m, qo_s
out = torch.empty(n_q, dtype=torch.int32, device=dev)
ix.lookup_var_dev(qd, qo, out)
torch.cuda.synchronize()
t0.record()
for _ in range(3):
    ix.lookup_var_dev(qd, qo, out)
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 3
hitrate = float((out != 0).float().mean().item())
ql = qo[1:] - qo[:-1]
res["mean_key_bytes"] = round(float(ql.float().mean().item()), 2)
res["frac_keys_at_most_17"] = round(float((ql <= 17).float().mean().item()), 3)
del ql
res["lookup_var_Glookups_per_s"] = round(n_q / ms / 1e6, 2)
res["hit_rate"] = round(hitrate, 3)
assert abs(hitrate - 0.7) < 0.01
assert bool((out[:nm] == taxa[:nm]).all())
res["var_batches_in_flight"] = int(os.environ.get("UMGAP_VAR_BATCHES", "1"))
res["inline_keys"] = os.environ.get("UMGAP_VAR_INLINE", "1") != "0"
# The queries above arrive in the order the members were inserted, so consecutive lookups verify against consecutive heap
# bytes (cache friendly).  Real reads do not: the same measurement over a RANDOM permutation of a 50 M-query sample.
n_s = min(n_q, 50_000_000)
lens = (qo[1:n_s + 1] - qo[:n_s])
perm = torch.randperm(n_s, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
# keep the 70 / 30 mix: sample positions across the whole query list
pick = (perm.to(torch.float64) * (n_q / n_s)).to(torch.int64)
plen = (qo[pick + 1] - qo[pick])
po = torch.zeros(n_s + 1, dtype=torch.int64, device=dev)
torch.cumsum(plen, 0, out=po[1:])
tot = int(po[-1].item())
src = torch.repeat_interleave(qo[pick] - po[:-1], plen) + torch.arange(tot, dtype=torch.int64, device=dev)
pd = qd[src]
del src, lens
pout = torch.empty(n_s, dtype=torch.int32, device=dev)
ix.lookup_var_dev(pd, po, pout)
torch.cuda.synchronize()
assert bool((pout == out[pick]).all())
t0.record()
for _ in range(3):
    ix.lookup_var_dev(pd, po, pout)
t1.record(); torch.cuda.synchronize()
res["lookup_var_shuffled_Glookups_per_s"] = round(n_s / (t0.elapsed_time(t1) / 3) / 1e6, 2)
del pd, po, pout, pick, perm, plen
if LENGTHS != "uniform":
    print(json.dumps(res))
    sys.exit(0)
# aggregation: reads of 16 hits drawn like the 9-mer reads (source path / unrelated / miss)
from umgap_amd.synth import SyntheticWorkload
del ix, qd, qo, out
torch.cuda.empty_cache()
w2 = SyntheticWorkload(n_tax=2_500_000, n_keys=64_000_000)
n_reads, stride = 10_000_000, 16
keys = w2.reads(0, n_reads, kmers_per_pair=stride)
hits = w2.index.lookup9_dev(keys)
dt = torch.empty(n_reads, dtype=torch.int32, device=dev); ds = torch.empty_like(dt)
for name, kw in (("mrtl_l1", dict(method="rmq", strategy="mrtl", lower_bound=1.0)), ("mrtl_l5", dict(method="rmq", strategy="mrtl", lower_bound=5.0)),
                 ("hybrid_f0.25_l1", dict(method="tree", strategy="hybrid", factor=0.25, lower_bound=1.0)),
                 ("rmq_hybrid_f0.25", dict(method="rmq", strategy="hybrid", factor=0.25, lower_bound=1.0)),
                 ("lca_star_l2", dict(method="tree", strategy="lca*", lower_bound=2.0))):
    w2.taxonomy.aggregate_strided_dev(hits, stride, n_reads, dt, ds, **kw)
    torch.cuda.synchronize()
    t0.record()
    for _ in range(3):
        w2.taxonomy.aggregate_strided_dev(hits, stride, n_reads, dt, ds, **kw)
    t1.record(); torch.cuda.synchronize()
    res["agg_%s_Mreads_per_s" % name] = round(n_reads / (t0.elapsed_time(t1) / 3) / 1e3, 1)
print(json.dumps(res))
