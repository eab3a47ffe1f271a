#!/usr/bin/env python3
"""The host-fed call a Rust / C++ host makes for the 9-mer pipeline (umgap_classify_records: protein records in HOST memory ->
one taxon per read in host memory; H2D, probe, seedextend, aggregate and D2H inside the call): the PCIe-inclusive rate of
surface B, beside the device-resident one of tools/bench_surface_b.py.  Reads = proteome fragments with 3 % of the residues changed,
2 x 50 residues per pair; the arrays are pinned (umgap_host_alloc / torch pinned memory) or pageable.  One JSON line.
    python tools/bench_records.py [n_proteome_fragments] [n_pairs]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from umgap_amd import _lib, synth
from umgap_amd.api import METHODS, STRATEGIES, Index, _dev_ptr, _np_ptr, _stream_ptr, check

L = _lib.load()
dev = torch.device("cuda", 0)
flen, F = 50, 2


def run(n_frag=50_000_000, n_pairs=10_000_000, steps=3):
    w = synth.SyntheticWorkload(n_tax=2_500_000, n_keys=1000, build_index=False)
    offs = torch.arange(0, (n_frag + 1) * flen, flen, dtype=torch.int64, device=dev)
    prot = torch.empty(n_frag * flen, dtype=torch.uint8, device=dev)
    _lib.check(L.umgap_synth_peptides_dev(0, n_frag, 7, _dev_ptr(offs), _dev_ptr(prot), 0, _stream_ptr(None)))
    taxa = w.d_taxon_of_row[torch.arange(n_frag, device=dev) % w.n_tax].contiguous()
    h = C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * (flen - 8), 0.5, 0, C.byref(h)))
    ix = Index(h.value)
    ix.bind(w.taxonomy)        # as `umgap prot2kmer2agg` runs it: the table bound to the taxonomy once at start-up
    del offs, taxa
    r = torch.empty(n_pairs * F * flen, dtype=torch.uint8, device=dev)
    _lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 44, 0, n_pairs, F, 30, _dev_ptr(r), 0, _stream_ptr(None)))
    torch.cuda.synchronize()
    res = {"pairs": n_pairs, "bytes_in_per_pair": F * flen + F * 8 + 8, "bytes_out_per_pair": 8, "index_keys": n_frag * (flen - 8),
           "pipeline": "prot2kmer2lca | seedextend -g1 -s3 | taxa2agg -l2 -a lca* on host arrays (umgap_classify_records), table bound to the taxonomy"}
    # the link alone: the same bytes copied from pinned memory, in 32 MB pieces on one stream and in one piece
    src = torch.empty(n_pairs * res["bytes_in_per_pair"], dtype=torch.uint8, pin_memory=True)
    dst = torch.empty_like(src, device=dev)
    for label, piece in (("h2d_pinned_32MB_pieces_GBps", 32 << 20), ("h2d_pinned_one_piece_GBps", src.numel())):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o in range(0, src.numel(), piece):
            dst[o:o + piece].copy_(src[o:o + piece], non_blocking=True)
        torch.cuda.synchronize()
        res[label] = round(src.numel() / (time.perf_counter() - t0) / 1e9, 2)
    del src, dst
    for name, pinned in (("pinned", True), ("pageable", False)):
        h_bytes = torch.empty(r.numel() + 16, dtype=torch.uint8, pin_memory=pinned)
        h_bytes[:r.numel()].copy_(r.cpu())
        h_off = torch.arange(0, (n_pairs * F + 1) * flen, flen, dtype=torch.int64)
        h_rf = torch.arange(0, n_pairs * F + 1, F, dtype=torch.int64)
        out_t = torch.empty(n_pairs, dtype=torch.int32)
        out_s = torch.empty(n_pairs, dtype=torch.int32)
        if pinned:
            h_off, h_rf, out_t, out_s = h_off.pin_memory(), h_rf.pin_memory(), out_t.pin_memory(), out_s.pin_memory()

        def call():
            check(L.umgap_classify_records(ix._h, w.taxonomy._h, METHODS["tree"], STRATEGIES["lca*"], 0.25, 2.0, 0, h_bytes.data_ptr(), h_off.data_ptr(), n_pairs * F,
                                           h_rf.data_ptr(), n_pairs, 1, 3, 1, out_t.data_ptr(), out_s.data_ptr()))
        call()
        times = []
        for _ in range(max(steps, 5)):       # every call timed on its own: min / median / max
            t0 = time.perf_counter()
            call()
            times.append(time.perf_counter() - t0)
        times.sort()
        dt = times[len(times) // 2]
        res[name] = {"s_per_call": round(dt, 4), "pairs_per_s": round(n_pairs / dt, 1), "GB_per_s_in": round(n_pairs * res["bytes_in_per_pair"] / dt / 1e9, 2),
                     "status_ok_frac": round(float((out_s == 0).float().mean()), 3), "calls": len(times), "route": int(L.umgap_records_last_route()),
                     "M_pairs_per_s_min_median_max": [round(n_pairs / times[-1] / 1e6, 1), round(n_pairs / dt / 1e6, 1), round(n_pairs / times[0] / 1e6, 1)]}
    ix.unbind(w.taxonomy)
    ix.close()
    return res


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*a)))
