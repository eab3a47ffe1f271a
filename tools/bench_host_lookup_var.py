#!/usr/bin/env python3
"""The host-fed get() of the tryptic index: umgap_index_lookup_var (peptides in HOST memory -> taxa in host memory, the link
included), from pinned and from pageable arrays.  One JSON line.
    python tools/bench_host_lookup_var.py [n_keys] [n_queries]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import _lib, synth
from umgap_amd.api import Index, _dev_ptr, _stream_ptr, check

L = _lib.load()
dev = torch.device("cuda", 0)


def peptides(first, n, seed):
    idx = torch.arange(first, first + n, dtype=torch.int64, device=dev)
    lens = 14 + (idx * 2654435761 + seed) % 32
    offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=offs[1:])
    total = int(offs[-1].item())
    data = torch.empty(total, dtype=torch.uint8, device=dev)
    check(L.umgap_synth_peptides_dev(first, n, seed, _dev_ptr(offs), _dev_ptr(data), 0, _stream_ptr(None)))
    return data, offs, total


def run(n_keys=200_000_000, n_q=50_000_000, steps=3):
    w = synth.SyntheticWorkload(n_tax=2_500_000, n_keys=1000, build_index=False)
    data, offs, total = peptides(0, n_keys, 43)
    taxa = w.d_taxon_of_row[torch.arange(n_keys, device=dev) % w.n_tax].contiguous()
    h = C.c_void_p()
    check(L.umgap_index_build_var_dev(_dev_ptr(data), _dev_ptr(offs), _dev_ptr(taxa), n_keys, total, 0.0, 0, C.byref(h)))
    ix = Index(h.value)
    # queries: the first 70 % of n_q are keys (hits), the rest peptides the index does not hold
    nh = n_q * 7 // 10
    qd, qo, qt = peptides(0, n_q, 43)
    md, mo, mt = peptides(n_keys + 5, n_q - nh, 43)
    cut = int(qo[nh].item())
    q_bytes = torch.cat([qd[:cut], md])
    q_offs = torch.cat([qo[:nh], mo + cut])
    n = q_offs.numel() - 1
    want = torch.empty(n, dtype=torch.int32, device=dev)
    ix.lookup_var_dev(q_bytes, q_offs, want, miss_value=0)
    torch.cuda.synchronize()
    want = want.cpu()
    res = {"index_keys": n_keys, "queries": n, "query_bytes": int(q_bytes.numel()), "hit_frac": round(float((want != 0).float().mean()), 3)}
    for name, pinned in (("pinned", True), ("pageable", False)):
        hb = torch.empty(q_bytes.numel() + 64, dtype=torch.uint8, pin_memory=pinned)
        hb[:q_bytes.numel()].copy_(q_bytes.cpu())
        ho = torch.empty(n + 1, dtype=torch.int64, pin_memory=pinned)
        ho.copy_(q_offs.cpu())
        out = torch.empty(n, dtype=torch.int32, pin_memory=pinned)

        def call():
            check(L.umgap_index_lookup_var(ix._h, hb.data_ptr(), ho.data_ptr(), n, 0, out.data_ptr()))
        call()
        assert torch.equal(out, want)
        t0 = time.perf_counter()
        for _ in range(steps):
            call()
        dt = (time.perf_counter() - t0) / steps
        res[name] = {"s_per_call": round(dt, 4), "G_lookups_per_s": round(n / dt / 1e9, 3), "GB_per_s_in": round((q_bytes.numel() + 8 * n) / dt / 1e9, 2)}
    return res


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*a)))
