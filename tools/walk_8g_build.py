#!/usr/bin/env python3
"""A walk-order index of the real 9-mer index's size class on ONE GPU: n_frag fragments of 50 residues (42 windows each; default
200 M = 8.4 G windows) through umgap_index_build_walk9_dev -- build time, sizes, free HBM at the end -- after the plain table of the
same proteome has answered a sample of reads (then freed: both do not fit together); the walk index must answer the same.
    python tools/walk_8g_build.py [n_frag] [load_factor]      -> one JSON line"""
import ctypes as C
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import _lib
from umgap_amd.api import Index, _dev_ptr, _stream_ptr, classify_prot9_dev
from umgap_amd.synth import SyntheticWorkload


def main():
    n_frag = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
    load = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    flen, per = 50, 42
    L = _lib.load()
    w = SyntheticWorkload(n_tax=200_000, n_keys=1000, build_index=False)
    offs = torch.arange(0, (n_frag + 1) * flen, flen, dtype=torch.int64, device="cuda")
    prot = torch.empty(n_frag * flen, dtype=torch.uint8, device="cuda")
    _lib.check(L.umgap_synth_peptides_dev(0, n_frag, 7, _dev_ptr(offs), _dev_ptr(prot), 0, _stream_ptr(None)))
    taxa = w.d_taxon_of_row[torch.arange(n_frag, device="cuda") % w.n_tax].contiguous()
    res = {"n_fragments": n_frag, "windows": n_frag * per, "load_factor": load}
    n_reads, F = 1_000_000, 2
    reads = torch.empty(n_reads * F * flen, dtype=torch.uint8, device="cuda")
    _lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 44, 0, n_reads, F, 30, _dev_ptr(reads), 0, _stream_ptr(None)))
    roffs = torch.arange(0, (n_reads * F + 1) * flen, flen, dtype=torch.int64, device="cuda")
    oo = torch.arange(0, (n_reads * F + 1) * per, per, dtype=torch.int64, device="cuda")

    def answers(ix, out, tag):
        h = torch.full((n_reads * F * per,), -3, dtype=torch.int32, device="cuda")
        ix.lookup_prot9_dev(reads, roffs, oo, h, miss_value=0)
        t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
        st = torch.full_like(t, -1)
        kw = dict(method="tree", strategy="lca*", lower_bound=2.0, seedextend=True, min_seed_size=3, max_gap_size=1)
        classify_prot9_dev(ix, w.taxonomy, reads, roffs, n_reads, F, t, st, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            classify_prot9_dev(ix, w.taxonomy, reads, roffs, n_reads, F, t, st, **kw)
        torch.cuda.synchronize()
        out[tag + "_fused_ms_per_1M_pairs"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        return h, t, st

    t0 = time.perf_counter()
    hp = C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * per, load, 0, C.byref(hp)))
    plain = Index(hp.value)
    torch.cuda.synchronize()
    res["plain_build_s"] = round(time.perf_counter() - t0, 1)
    res["plain_table_GB"] = round(plain.table_bytes / 1e9, 1)
    want = answers(plain, res, "plain")
    plain.close()
    gc.collect()
    torch.cuda.empty_cache()
    free0, total = torch.cuda.mem_get_info()
    res["free_GB_before_walk_build"] = round(free0 / 1e9, 1)
    t0 = time.perf_counter()
    hw = C.c_void_p()
    _lib.check(L.umgap_index_build_walk9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * per, load, 0, C.byref(hw)))
    walk = Index(hw.value)
    torch.cuda.synchronize()
    res["walk_build_s"] = round(time.perf_counter() - t0, 1)
    shift, entries, exc = walk.walk_info()
    res.update(walk_entries=entries, walk_shift=shift, walk_exceptions=exc, walk_index_GB=round(walk.table_bytes / 1e9, 1))
    free1, _ = torch.cuda.mem_get_info()
    res["free_GB_with_walk_index_resident"] = round(free1 / 1e9, 1)
    got = answers(walk, res, "walk")
    res["hits_identical_to_plain_table"] = bool(torch.equal(want[0], got[0]))
    res["hit_fraction"] = round(float((got[0] != 0).float().mean()), 3)
    res["fused_answers_identical"] = bool(torch.equal(want[1], got[1]) and torch.equal(want[2], got[2]))
    walk.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
