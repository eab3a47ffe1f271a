#!/usr/bin/env python3
"""Where the start-up of a device-backed `umgap` command goes: `umgap prot2kmer2agg` on an EMPTY input (index of 10.5 M 9-mers,
taxonomy of 100 k taxa: tools/end_to_end.py's files), five times, wall time and the command's own UMGAP_TIMING lines; and the
same for `umgap pept2lca` (index only) and `umgap taxa2agg` (taxonomy only).
    python tools/startup_times.py"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import end_to_end as E

work = "/tmp/umgap_startup"
os.makedirs(work, exist_ok=True)
idx, tsv, fasta, n_keys = E._make_inputs(work, 1000, 250_000)
res = {"index_keys": n_keys}
env = dict(os.environ, UMGAP_TIMING="1")
for name, cmd in (("prot2kmer2agg", [E.UMGAP, "prot2kmer2agg", "-l", "2", "-a", "lca*", idx, tsv]), ("pept2lca", [E.UMGAP, "pept2lca", idx]),
                  ("taxa2agg", [E.UMGAP, "taxa2agg", "-l", "2", "-a", "lca*", tsv]), ("uniq (no device)", [E.UMGAP, "uniq"])):
    ts, err = [], b""
    for _ in range(5):
        t0 = time.perf_counter()
        p = subprocess.run(cmd, input=b"", stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        ts.append(round(time.perf_counter() - t0, 3))
        err = p.stderr
    res[name] = {"wall_s": ts, "timing": [l for l in err.decode(errors="replace").splitlines() if "amdgpu.ids" not in l][-6:]}
print(json.dumps(res, indent=1))
