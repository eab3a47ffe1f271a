#!/usr/bin/env python3
"""Surface B of the 9-mer high-precision path: the reads arrive as protein
fragments (2 x 50 residues per pair, 100 B instead of 672 B of packed k-mers) and ONE kernel does
LDS-staged k-mer packing -> probe -> [seedextend -g1 -s3] -> taxa2agg -l2 -a lca*.
The index holds every 9-mer of a synthetic proteome (n_frag fragments of 50 residues); reads are proteome fragments with
`mutations_per_mille` of their residues replaced.  --walk builds the same index in walk order (umgap_index_build_walk9_dev) and
checks its answers bit for bit against the plain table's on the first batch.  Prints one JSON line.
    python tools/bench_surface_b.py [n_proteome_fragments] [n_pairs] [mutations_per_mille] [--walk]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import _lib, synth
from umgap_amd.api import Index, _dev_ptr, _stream_ptr, classify_prot9_dev

L = _lib.load()
dev = torch.device("cuda", 0)
flen, F = 50, 2
ALL_LEGS = (("classify_prot9", False, "tree", "lca*", 2.0), ("classify_prot9_seedextend", True, "tree", "lca*", 2.0),
            ("classify_prot9_seedextend_hybrid", True, "rmq", "hybrid", 1.0), ("classify_prot9_mrtl", False, "rmq", "mrtl", 1.0))


def run(n_frag=100_000_000, n_pairs=10_000_000, mut=30, walk=False, steps=5, workload=None, legs=ALL_LEGS):
    """Builds the proteome index (plain; with `walk` also the walk-order one, whose answers are checked bit for bit against the plain
    table's on the first batch), the read batches, and times the presets of scripts/umgap-analyse.sh:277-310.  Returns the dict."""
    w = workload if workload is not None else synth.SyntheticWorkload(n_tax=2_500_000, n_keys=1000, build_index=False)
    t0 = time.perf_counter()
    offs = torch.arange(0, (n_frag + 1) * flen, flen, dtype=torch.int64, device=dev)
    prot = torch.empty(n_frag * flen, dtype=torch.uint8, device=dev)
    _lib.check(L.umgap_synth_peptides_dev(0, n_frag, 7, _dev_ptr(offs), _dev_ptr(prot), 0, _stream_ptr(None)))
    taxa = w.d_taxon_of_row[torch.arange(n_frag, device=dev) % w.n_tax].contiguous()
    h = C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * (flen - 8), 0.5, 0, C.byref(h)))
    ix = Index(h.value)
    torch.cuda.synchronize()
    setup = time.perf_counter() - t0
    wx = None
    if walk:
        t0 = time.perf_counter()
        hw = C.c_void_p()
        _lib.check(L.umgap_index_build_walk9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * (flen - 8), 0.5, 0, C.byref(hw)))
        wx = Index(hw.value)
        torch.cuda.synchronize()
        walk_setup = time.perf_counter() - t0
    del offs, taxa
    batches = []
    for b in range(steps):
        r = torch.empty(n_pairs * F * flen, dtype=torch.uint8, device=dev)
        _lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 44, b * n_pairs, n_pairs, F, mut, _dev_ptr(r), 0, _stream_ptr(None)))
        batches.append(r)
    roffs = torch.arange(0, (n_pairs * F + 1) * flen, flen, dtype=torch.int64, device=dev)
    d_t = torch.empty(n_pairs, dtype=torch.int32, device=dev)
    d_s = torch.empty_like(d_t)
    res = {"n_proteome_fragments": n_frag, "index_keys": n_frag * (flen - 8), "table_GB": round(ix.table_bytes / 1e9, 2), "pairs_per_step": n_pairs,
           "input_bytes_per_pair": F * flen + (F * 8), "setup_s": round(setup, 1), "mutations_per_mille": mut}
    if walk:
        res.update({"layout": "walk order: table of window numbers + walk array (umgap_index_build_walk9_dev)", "walk_setup_s": round(walk_setup, 1),
                    "walk_table_GB": round(wx.table_bytes / 1e9, 2), "walk_array_GB": round(n_frag * (flen - 8) * 8 / 1e9, 2)})
        wx.bind(w.taxonomy)     # the walk array's taxa as interval codes (the reference answers below come from the plain, unbound table)
    else:
        ix.bind(w.taxonomy)     # values = preorder interval codes: the aggregate stage gathers nothing by id
    res["table_bound_to_taxonomy"] = True
    use = wx if walk else ix
    for name, se, method, strategy, lb in legs:
        kw = dict(method=method, strategy=strategy, factor=0.25, lower_bound=lb, seedextend=se, min_seed_size=3, max_gap_size=1)
        same = None
        if walk:
            t0_, s0_ = torch.empty_like(d_t), torch.empty_like(d_s)
            classify_prot9_dev(ix, w.taxonomy, batches[0], roffs, n_pairs, F, t0_, s0_, **kw)
        classify_prot9_dev(use, w.taxonomy, batches[0], roffs, n_pairs, F, d_t, d_s, **kw)
        torch.cuda.synchronize()
        if walk:
            same = bool(torch.equal(t0_, d_t) and torch.equal(s0_, d_s))
            del t0_, s0_
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(steps):
            classify_prot9_dev(use, w.taxonomy, batches[i], roffs, n_pairs, F, d_t, d_s, **kw)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / steps
        st = d_s.cpu()
        res[name] = {"pairs_per_s": round(n_pairs / ms * 1e3, 1), "ms_per_step": round(ms, 3), "G_lookups_per_s": round(n_pairs * 84 / ms / 1e6, 2),
                     "status_ok_frac": round(float((st == 0).float().mean()), 3), "status_empty_frac": round(float((st == 1).float().mean()), 3),
                     "status_unsupported_frac": round(float((st == 4).float().mean()), 5)}
        if same is not None:
            res[name]["bit_identical_to_plain_table"] = same
    # host-fed estimate: PCIe Gen5 x16 (~50 GB/s achievable) on 116 B per pair in + 8 B out
    res["pcie_bound_pairs_per_s_at_50GBps"] = round(50e9 / (res["input_bytes_per_pair"] + 8), 1)
    if walk:
        wx.unbind(w.taxonomy)
        wx.close()
    else:
        ix.unbind(w.taxonomy)
    ix.close()
    return res


if __name__ == "__main__":
    WALK = "--walk" in sys.argv
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    print(json.dumps(run(int(argv[0]) if len(argv) > 0 else 100_000_000, int(argv[1]) if len(argv) > 1 else 10_000_000, int(argv[2]) if len(argv) > 2 else 30, walk=WALK)))
