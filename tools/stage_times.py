#!/usr/bin/env python3
"""Which stage of the unchanged four-process chain (scripts/umgap-analyse.sh:299-304) sets its pace?  Runs the stages one after
another over files (each alone on the machine) and prints seconds, MB/s of input and pairs/s per stage.  Inputs as in
tools/end_to_end.py; --tryptic: the tryptic-sensitivity chain (:289-293) over a 50 M-key tryptic index instead.
    python tools/stage_times.py [n_pairs] [--tryptic]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import end_to_end as E

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_pairs = int(args[0]) if args else 2_000_000
work = "/tmp/umgap_stages"
os.makedirs(work, exist_ok=True)


def _make_tryptic_inputs(work, n_pairs, n_keys=50_000_000):
    """A tryptic index (tools/bench_tryptic_fused.py's peptides), a taxonomy TSV and protein FASTA of n_pairs pairs (2 records of 2
    peptides each): what FGSpp hands to `prot2tryp2lca` in the tryptic presets (scripts/umgap-analyse.sh:289-298)."""
    import ctypes as C

    import torch

    import bench_tryptic_fused as T
    from umgap_amd import _lib, synth
    from umgap_amd.api import Index, _dev_ptr
    ids, par, valid, rank = synth.synthetic_taxonomy(100_000, 140_000)
    tsv = os.path.join(work, "taxons.tsv")
    synth.write_taxonomy_tsv(tsv, ids, par, valid, rank)
    data, offs, total = T.write_peptides(None, 0, n_keys)
    taxa = torch.from_numpy(ids.view("int32")).to(T.dev)[torch.arange(n_keys, device=T.dev) % len(ids)].contiguous()
    h = C.c_void_p()
    _lib.check(T.L.umgap_index_build_var_dev(_dev_ptr(data), _dev_ptr(offs), _dev_ptr(taxa), n_keys, total, 0.0, 0, C.byref(h)))
    ix = Index(h.value)
    idx = os.path.join(work, "tryptic.umgapix")
    ix.save(idx)
    ix.close()
    del data, offs, taxa
    e = torch.arange(0, n_pairs * 4, dtype=torch.int64, device=T.dev)
    r = T.splitmix(e ^ 0x5DEECE66D)
    ids_r = torch.where((r & 0xFF) % 100 < 70, ((r >> 8) & 0x7FFFFFFFFFFF) % n_keys, 10**12 + ((r >> 8) & 0x7FFFFFFFFFFF)).contiguous()
    d, o, tot = T.write_peptides(ids_r, 0, ids_r.numel())
    host, ho = d.cpu().numpy(), o.cpu().numpy()
    fasta = os.path.join(work, "proteins_tryptic.fa")
    with open(fasta, "wb") as f:
        for rec in range(2 * n_pairs):
            f.write(b">read%d/%d\n" % (rec // 2, rec % 2 + 1))
            f.write(host[ho[2 * rec]:ho[2 * rec + 2]].tobytes())
            f.write(b"\n")
    torch.cuda.empty_cache()
    return idx, tsv, fasta


if "--tryptic" in sys.argv:
    idx, tsv, fasta = _make_tryptic_inputs(work, n_pairs)
    stages = [("prot2tryp2lca -l9 -L45", ["prot2tryp2lca", "-l9", "-L45", idx]), ("uniq -d /", ["uniq", "-d", "/"]),
              ("taxa2agg -l1 -m rmq -a mrtl", ["taxa2agg", "-l1", "-m", "rmq", "-a", "mrtl", tsv])]
else:
    idx, tsv, fasta, n_index_keys = E._make_inputs(work, n_pairs, 250_000)
    stages = [("prot2kmer2lca -o", ["prot2kmer2lca", "-o", idx]), ("seedextend -g1 -s3", ["seedextend", "-g1", "-s3"]), ("uniq -d /", ["uniq", "-d", "/"]),
              ("taxa2agg -l2 -a lca*", ["taxa2agg", "-l2", "-a", "lca*", tsv])]
res = {"pairs": n_pairs, "stages": []}
src = fasta
for i, (name, args) in enumerate(stages):
    dst = os.path.join(work, "stage%d.out" % i)
    # start-up (index load, taxonomy build) measured on an empty input and subtracted
    t0 = time.perf_counter()
    subprocess.run([E.UMGAP] + args, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, check=True)
    t_start = time.perf_counter() - t0
    import resource
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.perf_counter()
    with open(src, "rb") as fi, open(dst, "wb") as fo:
        subprocess.run([E.UMGAP] + args, stdin=fi, stdout=fo, check=True)
    dt = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    run = max(dt - t_start, 1e-9)
    res["stages"].append({"stage": name, "s": round(dt, 3), "start_up_s": round(t_start, 3), "cpu_user_s": round(r1.ru_utime - r0.ru_utime, 3),
                          "cpu_sys_s": round(r1.ru_stime - r0.ru_stime, 3), "in_MB": round(os.path.getsize(src) / 1e6, 1),
                          "out_MB": round(os.path.getsize(dst) / 1e6, 1), "in_MB_per_s": round(os.path.getsize(src) / 1e6 / run, 1),
                          "pairs_per_s": round(n_pairs / run, 1)})
    src = dst
# the same stages chained through pipes, as the script runs them
t0 = time.perf_counter()
procs = []
with open(fasta, "rb") as fi, open(os.path.join(work, "chain.out"), "wb") as fo:
    prev = fi
    for i, (name, a) in enumerate(stages):
        last = i == len(stages) - 1
        p = subprocess.Popen([E.UMGAP] + a, stdin=prev, stdout=fo if last else subprocess.PIPE)
        if prev is not fi:
            prev.close()
        prev = p.stdout
        procs.append(p)
    rcs = [p.wait() for p in procs]
dt = time.perf_counter() - t0
res["chain"] = {"s": round(dt, 3), "pairs_per_s": round(n_pairs / dt, 1), "exit": rcs,
                "same_output_as_staged": open(os.path.join(work, "chain.out"), "rb").read() == open(src, "rb").read()}
print(json.dumps(res))
