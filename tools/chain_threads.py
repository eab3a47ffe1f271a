#!/usr/bin/env python3
"""The unchanged four-process chain (tools/end_to_end.py) under different worker-thread counts per stage (UMGAP_THREADS): the four
stages share the box's cores, so 16 threads each is not obviously right.   python tools/chain_threads.py [n_pairs] [t1,t2,...]"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import end_to_end as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
threads = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 4, 6, 8, 12]
work = "/tmp/umgap_e2e"
os.makedirs(work, exist_ok=True)
idx, tsv, fasta, _ = E._make_inputs(work, n, 250_000)
res = {}
for t in threads:
    if t:
        os.environ["UMGAP_THREADS"] = str(t)
    else:
        os.environ.pop("UMGAP_THREADS", None)
    sock = os.path.join(work, "socket_t%d" % t)
    if os.path.exists(sock):
        os.unlink(sock)
    server = subprocess.Popen([E.UMGAP, "prot2kmer2lca", "-m", "-o", "-s", sock, idx], stdout=subprocess.DEVNULL)
    try:
        while not os.path.exists(sock):
            time.sleep(0.02)
        out = os.path.join(work, "out.t%d" % t)
        E._socket_pipeline(sock, E._head(fasta, 2000, os.path.join(work, "warm.fa")), tsv, out + ".warm")
        best = min(E._socket_pipeline(sock, fasta, tsv, out) for _ in range(2))
        res["threads_%s" % (t or "default")] = round(n / best / 1e6, 3)
    finally:
        server.terminate()
        server.wait()
print(json.dumps({"pairs": n, "chain_M_pairs_per_s": res}))
