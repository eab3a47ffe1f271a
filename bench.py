#!/usr/bin/env python3
"""bench.py — paired reads/s classified by the 9-mer high-precision hot path on MI355X.

One "step" = one pass of the hot path (packed 9-mers -> table probe -> per-pair LCA* with -l 2)
over one batch of synthetic read pairs whose keys are already resident in HBM.  Workload =
BASELINE.json configs[2]: 50 M synthetic 150-bp pairs (84 9-mers each) against the synthetic
9-mer index, processed as 5 resident batches of 10 M pairs (steps cycle over the batches).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: reads are independent units, so by default every rank holds a replica of the table and
classifies its own reads with no data-path collective (weak scaling).  The north star's sharded
mode (index sharded by bucket range, keys / hits exchanged with RCCL all-to-all over xGMI) is
then measured too and reported under "sharded" in the same line (`--mode sharded` makes it the
headline instead; `--mode replica` skips it).
Prints ONE JSON line on rank 0 (contract: see the driver's prompt; DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_LOOKUP = 76          # 64 B bucket + 8 B key + 4 B taxon          
KMERS = 84
BYTES_PER_PAIR = KMERS * BYTES_PER_LOOKUP + KMERS * 4 + 4   # 6724: + hits re-read by aggregate + result
HBM_PEAK_GBPS = 8000.0         # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["auto", "replica", "sharded"], default="auto")
    ap.add_argument("--pairs-per-step", type=int, default=int(os.environ.get("UMGAP_BENCH_PAIRS", 10_000_000)))
    ap.add_argument("--batches", type=int, default=5, help="resident batches (5 x 10 M = the 50 M-pair config)")
    ap.add_argument("--n-keys", type=int, default=int(os.environ.get("UMGAP_BENCH_KEYS", 8_000_000_000)))
    ap.add_argument("--n-tax", type=int, default=2_500_000)
    ap.add_argument("--load-factor", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--seedextend", action="store_true", help="insert seedextend -g1 -s3 per fragment between lookup and aggregate")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path (process group, barriers, sharded extra) even with one rank")
    ap.add_argument("--unfused", action="store_true", help="run lookup and aggregate as separate kernels (hits through HBM)")
    ap.add_argument("--no-bind", action="store_true", help="keep taxon ids in the table (the fused kernel then gathers {pre,last} by id)")
    return ap.parse_args()


def cpu_baseline(n_tax, ids, parents, valid, rank, taxonomy_handle, d_taxon_of_row, d_row_of_id, device):
    """Oracle ("port") timed on the host cores, bounded sample (DESIGN.md "CPU baseline")."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from umgap_amd import _lib, synth
    from umgap_amd.api import _dev_ptr, _stream_ptr
    cores = os.cpu_count() or 1
    n_cpu_keys, n_pairs = 32_000_000, 200_000
    # CPU index: the first n_cpu_keys members of the synthetic index as a sorted array (binary search =
    # ~25 dependent accesses per get, standing in for the fst automaton walk)
    i = np.arange(n_cpu_keys, dtype=np.uint64)
    x = (np.uint64(synth.MUL_A) * i + np.uint64(synth.ADD_B)) % np.uint64(synth.M20_9)
    codes = np.array([ord(c) - 65 for c in "ACDEFGHIKLMNPQRSTVWY"], np.uint64)
    keys = np.zeros(n_cpu_keys, np.uint64)
    for j in range(9):
        keys |= codes[(x % np.uint64(20)).astype(np.int64)] << np.uint64(5 * j)
        x //= np.uint64(20)
    vals = ids[(i % np.uint64(n_tax)).astype(np.int64)].astype(np.uint32)
    cpu_index = O.SortedIndex(keys, vals)
    del i, x, keys, vals
    # the sample: the same generator, bound to the CPU index size so the hit mix is the same
    d_keys = torch.empty(n_pairs * KMERS, dtype=torch.int64, device=device)
    _lib.check(_lib.load().umgap_synth_reads9_dev(taxonomy_handle, _dev_ptr(d_taxon_of_row), _dev_ptr(d_row_of_id), n_tax, n_cpu_keys,
                                                  synth.MUL_A, synth.ADD_B, synth.SEED_READS, 0, n_pairs, KMERS, _dev_ptr(d_keys), _stream_ptr(None)))
    torch.cuda.synchronize()
    q = d_keys.cpu().numpy().view(np.uint64)
    t0 = time.perf_counter()
    hits = cpu_index.lookup9(q, 0, nthreads=cores)       # lookup stage: all cores (rayon in the reference)
    t_lookup = time.perf_counter() - t0
    agg = O.Aggregator(O.Taxonomy.from_arrays(ids, parents, valid, rank), "tree", "lca*")
    offs = np.arange(n_pairs + 1, dtype=np.uint64) * KMERS
    t0 = time.perf_counter()
    agg.taxa2agg_check(hits, offs, lower_bound=2.0, nthreads=1)   # aggregate stage: one thread (taxa2agg)
    t_agg = time.perf_counter() - t0
    # the reference runs the stages as concurrent processes: throughput = slowest stage
    value = n_pairs / max(t_lookup, t_agg)
    import shutil
    have_ref = bool(shutil.which("cargo")) or bool(shutil.which("rustc"))
    return {"value": round(value, 1), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "restated CPU baseline (not the Rust binary: %s on this box); %d pairs (%d lookups) vs a %d-key sorted-array index; "
                      "lookup on %d threads %.2fs, LCA* -l2 on 1 thread %.2fs; pipelined stages => rate of the slower one" % (
                          "a Rust toolchain exists but the reference sources do not travel" if have_ref else "no cargo/rustc",
                          n_pairs, n_pairs * KMERS, n_cpu_keys, cores, t_lookup, t_agg)}


def sharded_extra(args, w, keys, rank, world, local_rank, device, B, d_hits, d_taxon, d_status, main_out):
    """The north star's sharded-index mode, measured after the headline run.  A watchdog prints the
    headline line and exits if a collective stalls, so this section can never cost the main result."""
    import threading

    import torch
    import torch.distributed as dist
    from umgap_amd import synth
    from umgap_amd.api import Index
    from umgap_amd.dist import GpuOps, ShardedLookup

    def bail():
        if rank == 0:
            main_out["sharded"] = {"error": "timed out after 180 s"}
            print(json.dumps(main_out), flush=True)
        os._exit(0)

    dog = threading.Timer(180.0, bail)
    dog.daemon = True
    dog.start()
    try:
        if w.index is not None:     # the replica has done its job: its HBM goes to the shard and the exchange buffers
            w.index.close()
            w.index = None
        shard = Index.synthetic(args.n_keys, w.d_taxon_of_row, synth.MUL_A, synth.ADD_B, args.load_factor, rank, world)
        sl = ShardedLookup(GpuOps(shard, device), rank, world, B * KMERS, chunks=4)
        steps = max(2, min(3, args.steps))

        def one(i):
            sl.lookup(keys[i % len(keys)], d_hits)
            w.taxonomy.aggregate_strided_dev(d_hits, KMERS, B, d_taxon, d_status, method="tree", strategy="lca*", lower_bound=2.0)

        one(0)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            one(i + 1)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        sl.check_overflow()
        st = d_status.cpu().numpy()
        assert ((st == 0) | (st == 1)).all()
        res = {"value": round(world * steps * B / float(el.item()), 1), "unit": "pairs/s", "steps": steps,
               "ms_per_step": round(1e3 * float(el.item()) / steps, 3), "shard_table_GB": round(shard.table_bytes / 1e9, 2),
               "exchange": "all_to_all_single: 8 B/key out, 4 B/key back, %d keys per rank per step, 4 pieces over 2 streams" % (B * KMERS)}
    except Exception as e:  # noqa: BLE001
        res = {"error": "%s: %s" % (type(e).__name__, e)}
    dog.cancel()
    return res


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    mode = args.mode
    if mode == "auto":
        mode = "replica"

    from umgap_amd import synth
    from umgap_amd.api import classify9_dev, seedextend_dev
    t_setup = time.perf_counter()
    sharded_mode = mode == "sharded" and world > 1
    w = synth.SyntheticWorkload(n_tax=args.n_tax, n_keys=args.n_keys, id_space=int(args.n_tax * 1.4), load_factor=args.load_factor,
                                device=local_rank, shard=rank if sharded_mode else 0, n_shards=world if sharded_mode else 1)
    torch.cuda.synchronize()
    B = args.pairs_per_step
    n_batches = max(1, min(args.batches, args.steps + args.warmup))
    # every rank owns different pairs: rank r, batch b -> pairs [(r*n_batches + b) * B, +B)
    keys = [w.reads((rank * n_batches + b) * B, B) for b in range(n_batches)]
    d_hits = torch.empty(B * KMERS, dtype=torch.int32, device=device)
    d_taxon = torch.empty(B, dtype=torch.int32, device=device)
    d_status = torch.empty(B, dtype=torch.int32, device=device)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    sharded = None
    if sharded_mode:
        from umgap_amd.dist import GpuOps, ShardedLookup
        sharded = ShardedLookup(GpuOps(w.index, device), rank, world, B * KMERS, chunks=4)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    fused = not args.unfused and not sharded_mode
    # the table is bound to the taxonomy for the fused LCA* kernel (values = preorder interval codes: no by-id gathers);
    # a one-time in-place pass, undone before the kernels that want taxon ids
    t_bind = None
    if fused and not args.no_bind:
        tb = time.perf_counter()
        w.index.bind(w.taxonomy)
        torch.cuda.synchronize()
        t_bind = time.perf_counter() - tb

    def step(i, timed_idx=None):
        k = keys[i % n_batches]
        if fused:
            if timed_idx is not None:
                ev[timed_idx][0].record()
            classify9_dev(w.index, w.taxonomy, k, B, KMERS, d_taxon, d_status, method="tree", lower_bound=2.0,
                          fragments_per_read=2 if args.seedextend else 0, min_seed_size=3, max_gap_size=1)
            if timed_idx is not None:
                ev[timed_idx][1].record()
            return
        if sharded is None:
            if timed_idx is not None:
                ev[timed_idx][0].record()
            w.index.lookup9_dev(k, d_hits, miss_value=0)
            if timed_idx is not None:
                ev[timed_idx][1].record()
        else:
            sharded.lookup(k, d_hits, ev[timed_idx] if timed_idx is not None else None)
        if args.seedextend:
            seedextend_dev(d_hits, stride=KMERS // 2, n_records=2 * B, min_seed_size=3, max_gap_size=1)
        w.taxonomy.aggregate_strided_dev(d_hits, KMERS, B, d_taxon, d_status, method="tree", strategy="lca*", lower_bound=2.0)

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, i)
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if sharded is not None:
        sharded.check_overflow()
    # sanity on the last batch: statuses are OK/EMPTY only and the answers are not degenerate
    st = d_status.cpu().numpy()
    assert ((st == 0) | (st == 1)).all(), "unexpected per-read status in the bench output"

    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)
    lookups = B * KMERS
    # fused: the one kernel carries the whole path (6 724 B per pair); unfused: the probe kernel (76 B per lookup)
    alg_bytes = B * BYTES_PER_PAIR if fused else lookups * BYTES_PER_LOOKUP
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    value = world * args.steps * B / elapsed

    out = {
        "metric": "paired reads/sec classified (9-mer high-precision)", "value": round(value, 1), "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 keys / u32 taxa (integer, no MFMA)",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[2]: synthetic 150 bp paired reads (84 packed 9-mers per pair), "
                               "%d resident batches x %d pairs per GPU; lookup%s + taxa2agg -l2 -a lca*%s" % (
                                   n_batches, B, " + seedextend -g1 -s3" if args.seedextend else "", " (one fused kernel)" if fused else ""),
                   "pairs_per_step_per_gpu": B, "n_keys": args.n_keys, "n_tax": args.n_tax,
                   "table_GB": round(w.index.table_bytes / 1e9, 2), "load_factor": args.load_factor,
                   "parallelism": ("1 GPU" if world == 1 else "%s x%d" % (mode, world)), "setup_s": round(t_setup, 1),
                   "table_bound_to_taxonomy": t_bind is not None, "bind_s": None if t_bind is None else round(t_bind, 3)},
        "roofline": {"bound": "hbm", "kernel": "k_classify9" if fused else "k_lookup9", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(kernel_ms, 3),
                     "whole_path_frac": round(value / world * BYTES_PER_PAIR / 1e9 / HBM_PEAK_GBPS, 4)},
    }
    # PMC-derived traffic, if a profile of this round has been committed (profiles/traffic.json)
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            if (tj.get("pairs_per_step") == B and tj.get("n_keys") == args.n_keys and tj.get("kernel") == out["roofline"]["kernel"]
                    and tj.get("load_factor", args.load_factor) == args.load_factor):
                out["roofline"]["traffic"] = tj["hbm_bytes_per_launch"]
        except Exception:
            pass
    if world == 1 and fused and not args.seedextend:
        # the full high-precision pipeline: seedextend -g1 -s3 per fragment between probe and aggregate
        # (scripts/umgap-analyse.sh:302), same batches, same kernel with the stage switched on
        def se_step(i):
            classify9_dev(w.index, w.taxonomy, keys[i % n_batches], B, KMERS, d_taxon, d_status, method="tree", lower_bound=2.0,
                          fragments_per_read=2, min_seed_size=3, max_gap_size=1)
        se_step(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            se_step(i + 1)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        out["with_seedextend"] = {"value": round(args.steps * B / el, 1), "unit": "pairs/s", "ms_per_step": round(1e3 * el / args.steps, 3),
                                  "pipeline": "probe -> seedextend -g1 -s3 (per read end) -> taxa2agg -l2 -a lca*, one fused kernel"}
    if world == 1 and fused and not args.seedextend:
        # the filter + aggregate stages of the two 9-mer sensitivity pipelines (scripts/umgap-analyse.sh:277-288) through the
        # fused kernel for the other strategies, on the same batches (the script feeds them 6-frame translations, ~500 k-mers
        # per pair, which go through umgap_classify_records); reported next to the headline, not part of it
        others = {}
        for name, kw in (("hybrid: seedextend -g1 -s3 | taxa2agg -l1 -a hybrid -f 0.25 (stages of umgap-analyse.sh:283-288 on the benchmark's 84-k-mer pairs)", dict(method="tree", strategy="hybrid", factor=0.25, lower_bound=1.0, min_seed_size=3)),
                         ("mrtl: seedextend -g1 -s2 | taxa2agg -l1 -m rmq -a mrtl (stages of umgap-analyse.sh:277-282 on the same pairs)", dict(method="rmq", strategy="mrtl", lower_bound=1.0, min_seed_size=2))):
            def o_step(i):
                classify9_dev(w.index, w.taxonomy, keys[i % n_batches], B, KMERS, d_taxon, d_status, fragments_per_read=2, max_gap_size=1, **kw)
            o_step(0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_o = min(args.steps, 5)
            for i in range(n_o):
                o_step(i + 1)
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            others[name] = {"value": round(n_o * B / el, 1), "unit": "pairs/s", "ms_per_step": round(1e3 * el / n_o, 3)}
        out["other_pipelines"] = others
    if w.index is not None and w.index.bound:
        w.index.unbind(w.taxonomy)          # what follows (sharded mode) wants taxon ids
    if rank == 0 and world == 1 and not args.no_gather:
        # measured random-sector gather ceilings over the same table (SURVEY.md §8d second denominator)
        n = 1 << 28
        d_out = torch.empty(n, dtype=torch.int32, device=device)
        ceil = {}
        for bpp, in_flight in ((64, 8), (128, 4)):   # pure random reads, several in flight per lane like the probe
            code = bpp | (in_flight << 16)
            w.index.bench_gather_dev(n, code, d_out)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            w.index.bench_gather_dev(n, code, d_out)
            b.record()
            torch.cuda.synchronize()
            ceil[str(bpp)] = round(n * bpp / (a.elapsed_time(b) * 1e-3) / 1e9, 1)
        del d_out
        out["roofline"]["measured_gather_GBps"] = ceil
        out["roofline"]["frac_of_measured_64B_gather"] = round(lookups * 64 / (kernel_ms * 1e-3) / 1e9 / ceil["64"], 4)
    if multi and args.mode == "auto":
        out["sharded"] = sharded_extra(args, w, keys, rank, world, local_rank, device, B, d_hits, d_taxon, d_status, out)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.n_tax, w.ids, w.parents, w.valid, w.rank, w.taxonomy._h, w.d_taxon_of_row, w.d_row_of_id, device)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
