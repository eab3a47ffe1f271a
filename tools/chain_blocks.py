#!/usr/bin/env python3
"""The unchanged four-process chain through the resident server under different block sizes (UMGAP_BLOCK_BYTES): a stage tokenizes a
block only when it has all of it, so four stages of 64 MB blocks hold a quarter of a second of stream between them before the
first read comes out.   python tools/chain_blocks.py [n_pairs] [MB,MB,...]"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import end_to_end as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 32, 16, 8]
work = "/tmp/umgap_e2e"
os.makedirs(work, exist_ok=True)
idx, tsv, fasta, _ = E._make_inputs(work, n, 250_000)
res = {}
for mb in sizes:
    os.environ["UMGAP_BLOCK_BYTES"] = str(mb << 20)
    sock = os.path.join(work, "socket_b%d" % mb)
    if os.path.exists(sock):
        os.unlink(sock)
    server = subprocess.Popen([E.UMGAP, "prot2kmer2lca", "-m", "-o", "-s", sock, idx], stdout=subprocess.DEVNULL)
    try:
        while not os.path.exists(sock):
            time.sleep(0.02)
        out = os.path.join(work, "out.b%d" % mb)
        E._socket_pipeline(sock, E._head(fasta, 2000, os.path.join(work, "warm.fa")), tsv, out + ".warm")
        best = min(E._socket_pipeline(sock, fasta, tsv, out) for _ in range(3))
        res["block_%d_MB" % mb] = round(n / best / 1e6, 3)
    finally:
        server.terminate()
        server.wait()
print(json.dumps({"pairs": n, "chain_M_pairs_per_s": res}))
