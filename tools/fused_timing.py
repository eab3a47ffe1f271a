#!/usr/bin/env python3
"""Phase times of the fused `umgap prot2kmer2agg` (UMGAP_TIMING=1) on the end-to-end input: python tools/fused_timing.py [n_pairs] [threads,threads,...]
(the second argument runs the default pipeline once per UMGAP_THREADS value instead)"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import end_to_end as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
work = "/tmp/umgap_e2e"
os.makedirs(work, exist_ok=True)
idx, tsv, fasta, _ = E._make_inputs(work, n, 250_000)
env = dict(os.environ, UMGAP_TIMING="1")
if len(sys.argv) > 2:
    for th in sys.argv[2].split(","):
        env["UMGAP_THREADS"] = th
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            with open(fasta, "rb") as fi, open(os.path.join(work, "out.fused"), "wb") as fo:
                p = subprocess.run([E.UMGAP, "prot2kmer2agg", "-g", "1", "-s", "3", "-d", "/", "-l", "2", "-a", "lca*", idx, tsv], stdin=fi, stdout=fo, stderr=subprocess.PIPE, env=env, check=True)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, " | ".join(p.stderr.decode().strip().splitlines()[-2:]))
        print("UMGAP_THREADS=%s: %d pairs in %.3f s = %.2f M pairs/s | %s" % (th, n, best[0], n / best[0] / 1e6, best[1]))
    sys.exit(0)
for extra in ([], ["--no-seedextend"]):
    for _ in range(2):
        t0 = time.perf_counter()
        with open(fasta, "rb") as fi, open(os.path.join(work, "out.fused"), "wb") as fo:
            p = subprocess.run([E.UMGAP, "prot2kmer2agg", "-g", "1", "-s", "3", "-d", "/", "-l", "2", "-a", "lca*"] + extra + [idx, tsv], stdin=fi, stdout=fo, stderr=subprocess.PIPE, env=env, check=True)
        dt = time.perf_counter() - t0
    print("%s: %d pairs in %.3f s = %.2f M pairs/s (%.0f MB in)" % (" ".join(extra) or "default", n, dt, n / dt / 1e6, os.path.getsize(fasta) / 1e6))
    print(p.stderr.decode().strip())
