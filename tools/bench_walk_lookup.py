#!/usr/bin/env python3
"""What a table laid out in WALK order buys on reads whose 9-mer windows overlap (index_walk.hip).
Index = all windows of a synthetic proteome (as tools/bench_surface_b.py); reads = proteome fragments with `mutations_per_mille` of
their residues replaced.  Compares umgap_index_lookup_prot9_dev -- the hit of every window -- over
  * the plain hash table (umgap_index_build_windows9_dev): one 64-byte request per window, and
  * the walk-order index (umgap_index_build_walk9_dev): every 8th window is probed, the windows behind it are read from the walk
    array and verified by induction (exact), the rest probed again,
bit for bit and in time; run under rocprofv3 --pmc TCC_EA0_RDREQ_sum it gives the requests of either kernel.
    python tools/bench_walk_lookup.py [n_proteome_fragments] [n_pairs] [mutations_per_mille]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import _lib, synth
from umgap_amd.api import Index, _dev_ptr, _stream_ptr

n_frag = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
mut = int(sys.argv[3]) if len(sys.argv) > 3 else 30
flen, F = 50, 2
W = flen - 8
dev = torch.device("cuda", 0)
L = _lib.load()
vp = C.c_void_p

ids, par, valid, rank = synth.synthetic_taxonomy(100_000, 140_000)
offs = torch.arange(0, (n_frag + 1) * flen, flen, dtype=torch.int64, device=dev)
prot = torch.empty(n_frag * flen, dtype=torch.uint8, device=dev)
_lib.check(L.umgap_synth_peptides_dev(0, n_frag, 7, _dev_ptr(offs), _dev_ptr(prot), 0, _stream_ptr(None)))
taxa = torch.from_numpy(ids.view("int32")).to(dev)[torch.arange(n_frag, device=dev) % len(ids)].contiguous()
h = vp()
_lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * W, 0.5, 0, C.byref(h)))
t1 = Index(h.value)
hw = vp()
_lib.check(L.umgap_index_build_walk9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * W, 0.5, 0, C.byref(hw)))
wx = Index(hw.value)
torch.cuda.synchronize()
res = {"proteome_fragments": n_frag, "index_keys": t1.num_keys, "plain_table_GB": round(t1.table_bytes / 1e9, 1), "position_table_GB": round(wx.table_bytes / 1e9, 1),
       "walk_array_GB": round(n_frag * W * 8 / 1e9, 1), "pairs": n_pairs, "windows": n_pairs * F * W, "mutations_per_mille": mut}
n_rf = n_pairs * F
reads = torch.empty(n_rf * flen, dtype=torch.uint8, device=dev)
_lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 44, 0, n_pairs, F, mut, _dev_ptr(reads), 0, _stream_ptr(None)))
roffs = torch.arange(0, (n_rf + 1) * flen, flen, dtype=torch.int64, device=dev)
ooffs = torch.arange(0, (n_rf + 1) * W, W, dtype=torch.int64, device=dev)
want = torch.empty(n_rf * W, dtype=torch.int32, device=dev)
got = torch.full_like(want, -1)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


ms_plain = timed(lambda: t1.lookup_prot9_dev(reads, roffs, ooffs, want, miss_value=0))
ms_walk = timed(lambda: wx.lookup_prot9_dev(reads, roffs, ooffs, got, miss_value=0))
nw = n_rf * W
res.update({"plain_ms": round(ms_plain, 3), "plain_G_windows_per_s": round(nw / ms_plain / 1e6, 1), "walk_ms": round(ms_walk, 3),
            "walk_G_windows_per_s": round(nw / ms_walk / 1e6, 1), "bit_identical": bool(torch.equal(want, got)),
            "hit_rate": round(float((want != 0).float().mean().item()), 3)})
print(json.dumps(res))
