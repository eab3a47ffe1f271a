#!/usr/bin/env python3
"""Start-up cost of a resident index: umgap_index_open of a UMGAPIX1 file (page cache warm) with 1 loader
thread and with the default 8.  Prints one JSON line."""
import json
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import synth
from umgap_amd.api import Index

n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/umgap_open"
os.makedirs(work, exist_ok=True)
free = shutil.disk_usage(work).free
w = synth.SyntheticWorkload(n_tax=100_000, n_keys=n_keys)
path = os.path.join(work, "big.umgapix")
res = {"n_keys": n_keys, "table_GB": round(w.index.table_bytes / 1e9, 2), "disk_free_GB": round(free / 1e9, 1)}
assert free > w.index.table_bytes * 1.2, "not enough scratch space"
t0 = time.perf_counter()
w.index.save(path)
res["save_s"] = round(time.perf_counter() - t0, 2)
keys = w.reads(0, 100_000)
want = w.index.lookup9_dev(keys).cpu()
del w
torch.cuda.empty_cache()
for threads in (1, 8, 16):
    os.environ["UMGAP_LOAD_THREADS"] = str(threads)
    t0 = time.perf_counter()
    ix = Index.open(path)
    dt = time.perf_counter() - t0
    res["open_%d_threads_s" % threads] = round(dt, 2)
    res["open_%d_threads_GBps" % threads] = round(res["table_GB"] / dt, 2)
    assert bool((ix.lookup9_dev(keys).cpu() == want).all())
    del ix
    torch.cuda.empty_cache()
os.remove(path)
print(json.dumps(res))
