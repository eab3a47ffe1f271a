#!/usr/bin/env python3
"""What does a STREAM beside the random reads cost at the request ceiling?  840 M random 64-byte quad probes of the benchmark's
128 GB table (8 in flight per lane, nothing stored per probe: ~50 G probes/s), plus -- per probe -- R bytes read from and W bytes
written to a side buffer, contiguous per wave step like the key and value streams of the look-up and exchange kernels.  From the
times: the price of a streamed 64-byte read and of a streamed 64-byte write in units of one random 64-byte read (the model behind
DESIGN.md section 7: why a sharded step that must WRITE 9 bytes per k-mer cannot run at the replica's pace).  One JSON line.
    python tools/gather_streams.py [n_keys] [n_probes]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import _lib, synth
from umgap_amd.api import Index, _dev_ptr, _stream_ptr

L = _lib.load()
fn = L.umgap_bench_gather_rw_dev
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]


def run(n_keys=8_000_000_000, n=840_000_000, reps=3, legs=None):
    dev = torch.device("cuda", 0)
    w = synth.SyntheticWorkload(n_tax=100_000, n_keys=1000, build_index=False)
    ix = Index.synthetic(n_keys, w.d_taxon_of_row, synth.MUL_A, synth.ADD_B, 0.5, 0, 1)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    side_in = torch.zeros(n * 16 + (64 << 20), dtype=torch.uint8, device=dev)     # (the grid's last step runs past n: the kernel stops its streams at the buffers' end)
    side_out = torch.empty(n * 16 + (64 << 20), dtype=torch.uint8, device=dev)
    res = {"table_GB": round(ix.table_bytes / 1e9, 1), "probes": n, "legs": {}}
    base = None
    for r, wr in (legs or ((0, 0), (4, 0), (8, 0), (16, 0), (0, 4), (0, 8), (0, 16), (8, 4), (8, 8), (16, 16), (0, 0x104), (0, 0x110), (0, 0x204))):
        best = None
        for _ in range(reps + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(fn(ix._h, n, r, wr, _dev_ptr(side_in), _dev_ptr(side_out), side_in.numel(), _dev_ptr(out), _stream_ptr(None)))
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None or ms < best else best
        if base is None:
            base = best
        leg = {"ms": round(best, 3), "G_probes_per_s": round(n / best / 1e6, 2)}
        if r or wr:
            # extra time over the pure gather, as random-read requests, per 64 bytes streamed
            extra_req = (best - base) / base * n               # in units of one random read
            leg["cost_per_64B_streamed_in_random_reads"] = round(extra_req / (n * (r + wr) / 64.0), 2)
        if wr >= 0x300:
            names = {1: "stores with sc1", 2: "stores with nt", 3: "stores with sc0 sc1", 4: "stores with sc0 sc1 nt",
                     8: "values through LDS, ONE wave of the workgroup stores its 2 KB (three of four waves never store)",
                     17: "2 steps' values stored together", 18: "4 steps' values stored together", 19: "8 steps' values stored together"}
            leg["cost_per_64B_streamed_in_random_reads"] = round((best - base) / base * n / (n * 4 / 64.0), 2)
            res["legs"]["write 4 B per probe, " + names[wr - 0x300]] = leg
            continue
        if wr >= 0x200:
            leg["cost_per_64B_streamed_in_random_reads"] = round((best - base) / base * n / (n * (wr - 0x200) / 64.0), 2)
            res["legs"]["write %d B per probe as one no-return 8-byte atomic exchange per lane and step" % (wr - 0x200)] = leg
            continue
        if wr >= 0x100:
            leg["cost_per_64B_streamed_in_random_reads"] = round((best - base) / base * n / (n * (wr - 0x100) / 64.0), 2)
            res["legs"]["write %d B per probe, the store issued behind the next step's loads (vmcnt excludes it)" % (wr - 0x100)] = leg
            continue
        res["legs"]["read %d B + write %d B per probe" % (r, wr)] = leg
    ix.close()
    return res


if __name__ == "__main__":
    if "--pmc" in sys.argv:     # under rocprofv3 --pmc: four legs, one launch each after a warm-up (tools/pmc_streams.sh)
        print(json.dumps(run(reps=1, legs=((0, 0), (16, 0), (0, 4), (0, 16)))))
    elif "--pmc5" in sys.argv:  # round 5: the pure gather, the 4-byte write stream, its LDS-staged one-writer form and the 8-step form
        print(json.dumps(run(reps=1, legs=((0, 0), (0, 4), (0, 0x308), (0, 0x313)))))
    elif "--variants" in sys.argv:
        print(json.dumps(run(legs=((0, 0), (0, 4), (0, 0x301), (0, 0x302), (0, 0x303), (0, 0x304), (0, 0x308), (0, 0x311), (0, 0x312), (0, 0x313), (0, 0)))))
    else:
        a = [int(x) for x in sys.argv[1:3]]
        print(json.dumps(run(*a)))
