#!/usr/bin/env python3
"""Start-up cost of the taxonomy: `umgap taxa2agg` on empty input with an NCBI-sized taxons.tsv (2.5 M rows)."""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umgap_amd.synth import synthetic_taxonomy, write_taxonomy_tsv
UMGAP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "umgap_amd", "bin", "umgap")
ids, par, valid, rank = synthetic_taxonomy()
tsv = "/tmp/taxons_full.tsv"
write_taxonomy_tsv(tsv, ids, par, valid, rank)
res = {"rows": len(ids), "tsv_MB": round(os.path.getsize(tsv) / 1e6, 1)}
for name, cmd in (("taxa2agg_lca", [UMGAP, "taxa2agg", "-a", "lca*", tsv]), ("taxa2agg_rmq_mrtl", [UMGAP, "taxa2agg", "-m", "rmq", "-a", "mrtl", tsv]), ("version_only", [UMGAP, "-V"])):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        pr = subprocess.run(cmd, input=b"", stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True, env=dict(os.environ, UMGAP_TIMING="1"))
        best = min(best, time.perf_counter() - t0)
        if pr.stderr.strip():
            res[name + "_phases"] = pr.stderr.decode().strip().splitlines()[-1]
    res[name + "_s"] = round(best, 3)
print(json.dumps(res))
