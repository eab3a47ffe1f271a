#!/usr/bin/env python3
"""A/B of umgap_classify_tryp_dev's chunking (UMGAP_TRYP_CHUNKS = 1, 2, 4, 8, ...): one index, one set of batches, every setting timed
call by call (median of 2 x steps calls) in the MRTL -l1 and tree-hybrid presets on the table bound to the taxonomy, answers compared
bit for bit with the unchunked call's.  Prints one JSON line.
    python tools/ab_tryp_chunks.py [n_keys] [n_pairs] [settings, comma separated] [environment variable to sweep = UMGAP_TRYP_CHUNKS]
A setting may name several variables: "512:UMGAP_AGG_GRID=10" sets the swept variable to 512 and UMGAP_AGG_GRID to 10."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from umgap_amd import synth
from umgap_amd.api import classify_tryp_dev

F, PEPS = 2, 2


def main():
    n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
    n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    settings = (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8,1,4").split(",")
    var = sys.argv[4] if len(sys.argv) > 4 else "UMGAP_TRYP_CHUNKS"
    steps = 3
    w = synth.SyntheticWorkload(n_tax=2_500_000, n_keys=1000, build_index=False)
    tw = synth.TrypticWorkload(n_keys, w, F, PEPS)
    ix, _ = tw.build_index()
    ix.bind(w.taxonomy)
    batches = [tw.batch(b, n_pairs) for b in range(steps)]
    d_t = torch.empty(n_pairs, dtype=torch.int32, device=w.device)
    d_s = torch.empty_like(d_t)
    presets = {"mrtl_l1": dict(method="rmq", strategy="mrtl", lower_bound=1.0, min_len=9, max_len=45),
               "tree_hybrid_f0.25_l1": dict(method="tree", strategy="hybrid", factor=0.25, lower_bound=1.0, min_len=9, max_len=45),
               "lca_star_l1": dict(method="tree", strategy="lca*", lower_bound=1.0, min_len=9, max_len=45)}
    res = {"n_keys": n_keys, "pairs_per_step": n_pairs, "runs": []}
    want = {}
    for chunks in settings:
        parts = str(chunks).split(":")
        os.environ[var] = parts[0]
        extra = [p.split("=") for p in parts[1:]]
        for k, v in extra:
            os.environ[k] = v
        row = {"setting": chunks, "variable": var}
        for name, kw in presets.items():
            per_step, same = [], True
            for i in range(2 * steps + 1):
                b = i % steps
                t0 = time.perf_counter()
                n_pieces = classify_tryp_dev(ix, w.taxonomy, batches[b][0], batches[b][1], n_pairs, F, d_t, d_s, **kw)
                torch.cuda.synchronize()
                if i:
                    per_step.append((time.perf_counter() - t0) * 1e3)
                key = (name, b)
                if key not in want:
                    want[key] = (d_t.clone(), d_s.clone(), n_pieces)
                else:
                    same = same and n_pieces == want[key][2] and torch.equal(d_t, want[key][0]) and torch.equal(d_s, want[key][1])
            row[name] = {"ms_median": round(sorted(per_step)[len(per_step) // 2], 3), "ms_min": round(min(per_step), 3), "same_answers_as_first_setting": same}
        for k, _ in extra:
            del os.environ[k]
        res["runs"].append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    ix.unbind(w.taxonomy)
    ix.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
