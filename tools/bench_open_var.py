#!/usr/bin/env python3
"""Start-up cost of a resident tryptic index: umgap_index_open of a UMGAPIX1 kind-2 file (page cache warm), 1 and 8 loader threads.
    python tools/bench_open_var.py [n_keys]"""
import ctypes as C
import json
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import _lib
from umgap_amd.api import Index, _dev_ptr, _stream_ptr

n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
work = "/tmp/umgap_open"
os.makedirs(work, exist_ok=True)
dev = torch.device("cuda", 0)
L = _lib.load()
lens = 14 + (torch.arange(n_keys, device=dev, dtype=torch.int64) * 2654435761 % 32)
offs = torch.zeros(n_keys + 1, dtype=torch.int64, device=dev)
torch.cumsum(lens, 0, out=offs[1:])
total = int(offs[-1].item())
data = torch.empty(total, dtype=torch.uint8, device=dev)
_lib.check(L.umgap_synth_peptides_dev(0, n_keys, 43, _dev_ptr(offs), _dev_ptr(data), 0, _stream_ptr(None)))
taxa = (torch.arange(n_keys, device=dev, dtype=torch.int64) % 1_000_003 + 1).to(torch.int32)
h = C.c_void_p()
_lib.check(L.umgap_index_build_var_dev(_dev_ptr(data), _dev_ptr(offs), _dev_ptr(taxa), n_keys, total, 0.0, 0, C.byref(h)))
ix = Index(h.value)
res = {"n_keys": n_keys, "layout": ix.var_layout, "table_GB": round(ix.table_bytes / 1e9, 2), "disk_free_GB": round(shutil.disk_usage(work).free / 1e9, 1)}
path = os.path.join(work, "tryptic.umgapix")
t0 = time.perf_counter()
ix.save(path)
res["save_s"] = round(time.perf_counter() - t0, 2)
nq = 1_000_000
qo = offs[:nq + 1].clone()
qd = data[:int(qo[-1].item())].clone()
out = torch.empty(nq, dtype=torch.int32, device=dev)
ix.lookup_var_dev(qd, qo, out)
want = out.cpu()
del ix, data, offs, taxa, lens
torch.cuda.empty_cache()
for threads in (1, 8):
    os.environ["UMGAP_LOAD_THREADS"] = str(threads)
    t0 = time.perf_counter()
    ix = Index.open(path)
    dt = time.perf_counter() - t0
    res["open_%d_threads_s" % threads] = round(dt, 2)
    res["open_%d_threads_GBps" % threads] = round(res["table_GB"] / dt, 2)
    ix.lookup_var_dev(qd, qo, out)
    assert bool((out.cpu() == want).all())
    del ix
    torch.cuda.empty_cache()
os.remove(path)
print(json.dumps(res))
