#!/usr/bin/env python3
"""The sharded step (umgap_sharded_classify9_dev) at world 1 inside ONE process, for profilers: bench.py runs its sharded
section in a child process, which rocprofv3 does not follow.  RCCL communicator of one rank (the exchange is a no-op; with
UMGAP_SHARD_NO_LOCAL=1 every key still goes through pack -> probe -> gather, which is what the other ranks' keys cost at N > 1).
`--pieces 1` serialises the stages (pack, probe, aggregate one after another over the whole batch): their stand-alone times.

    python tools/bench_sharded.py [--pieces 8] [--pairs 10000000] [--n-keys 8000000000] [--strategy lca*|hybrid|mrtl] [--steps 4]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pieces", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=10_000_000)
    ap.add_argument("--n-keys", type=int, default=8_000_000_000)
    ap.add_argument("--n-tax", type=int, default=2_500_000)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--strategy", default="lca*")
    ap.add_argument("--transport", choices=["rccl", "local"], default="rccl")
    args = ap.parse_args()
    import torch

    from umgap_amd import synth
    from umgap_amd.api import Index, classify9_dev
    from umgap_amd.dist import Comm, Sharded
    K = synth.KMERS_PER_PAIR
    w = synth.SyntheticWorkload(n_tax=args.n_tax, n_keys=args.n_keys, id_space=int(args.n_tax * 1.4), build_index=False)
    shard = Index.synthetic(args.n_keys, w.d_taxon_of_row, synth.MUL_A, synth.ADD_B, 0.5, 0, 1)
    B = args.pairs
    keys = [w.reads(b * B, B) for b in range(2)]
    comm = Comm.init_rank(Comm.unique_id(), 1, 0, 0) if args.transport == "rccl" else Comm.local(1)[0]
    per_piece = -(-B // args.pieces)
    sh = Sharded(shard, w.taxonomy, comm, per_piece * args.pieces * K, pieces=args.pieces)
    method = "rmq" if args.strategy == "mrtl" else "tree"
    lb = 2.0 if args.strategy == "lca*" else 1.0
    t = torch.empty(B, dtype=torch.int32, device="cuda")
    s = torch.empty_like(t)
    t0, s0 = torch.empty_like(t), torch.empty_like(t)
    classify9_dev(shard, w.taxonomy, keys[0], B, K, t0, s0, method=method, strategy=args.strategy, factor=0.25, lower_bound=lb)
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    for i in range(3):
        classify9_dev(shard, w.taxonomy, keys[i % 2], B, K, t0, s0, method=method, strategy=args.strategy, factor=0.25, lower_bound=lb)
    torch.cuda.synchronize()
    whole_ms = (time.perf_counter() - c0) / 3 * 1e3
    classify9_dev(shard, w.taxonomy, keys[0], B, K, t0, s0, method=method, strategy=args.strategy, factor=0.25, lower_bound=lb)
    sh.classify9(keys[0], B, K, t, s, method=method, strategy=args.strategy, factor=0.25, lower_bound=lb)
    torch.cuda.synchronize()
    sh.check()
    same = bool(torch.equal(t, t0) and torch.equal(s, s0))
    c0 = time.perf_counter()
    for i in range(args.steps):
        sh.classify9(keys[(i + 1) % 2], B, K, t, s, method=method, strategy=args.strategy, factor=0.25, lower_bound=lb)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - c0) / args.steps * 1e3
    sh.check()
    print(json.dumps({"sharded_ms_per_step": round(ms, 3), "whole_table_ms_per_step": round(whole_ms, 3), "pieces": args.pieces, "pairs": B,
                      "strategy": args.strategy, "own_keys": "packed" if os.environ.get("UMGAP_SHARD_NO_LOCAL") == "1" else "probed in place",
                      "bit_identical_to_whole_table": same, "transport": args.transport}))
    sh.close()
    comm.close()


if __name__ == "__main__":
    main()
