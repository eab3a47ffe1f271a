#!/usr/bin/env python3
"""bench.py — paired reads/s classified by the 9-mer high-precision hot path on MI355X.

One "step" = one pass of the hot path (packed 9-mers -> table probe -> per-pair LCA* with -l 2)
over one batch of synthetic read pairs whose keys are already resident in HBM.  Workload =
BASELINE.json configs[2]: 50 M synthetic 150-bp pairs (84 9-mers each) against the synthetic
9-mer index, processed as 5 resident batches of 10 M pairs (steps cycle over the batches).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without RANK in the environment starts its N ranks itself (one process per
GPU, rendezvous on 127.0.0.1) BEFORE anything touches the GPU and exits with their status.

N > 1 (or --force-dist): reads are independent units, so the headline `value` is the replica mode --
every rank holds the table and classifies its own reads with no data-path collective (weak scaling).
The north star's sharded mode (index sharded by bucket range, k-mers and hits exchanged over RCCL inside
libumgap_hip.so: umgap_sharded_classify9_dev) is measured in the same run and reported under "sharded"
(`--mode sharded` makes it the headline, `--mode replica` skips it).  "ranks_seen" comes from an RCCL
all-reduce.  Prints ONE JSON line on rank 0 (contract: the driver's prompt; DESIGN.md "Measurement").
`--rehearse-cpu` runs the launch / rendezvous / reduction plumbing on CPU over gloo (CI).
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
BYTES_PER_PAIR_FUSED = KMERS * (64 + 8) + 4                 # 6052: what the fused kernel must move (hits never leave the CU)
HBM_PEAK_GBPS = 8000.0         # MI355X HBM3E spec (MI355X_MICROARCH.md)
METRIC = "paired reads/sec classified (9-mer high-precision)"


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
    ap.add_argument("--pieces", type=int, default=8, help="pipelined pieces of a sharded step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-sharded", action="store_true", help="one GPU: skip the sharded leg (a child process with a process group of one over RCCL)")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no seedextend / other pipelines / workload sweep / end-to-end legs")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="keep roofline.traffic from profiles/traffic.json instead of two child runs under rocprofv3 --pmc")
    ap.add_argument("--seedextend", action="store_true", help="insert seedextend -g1 -s3 per fragment between lookup and aggregate")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path (process group, barriers, sharded extra) even with one rank")
    ap.add_argument("--unfused", action="store_true", help="run lookup and aggregate as separate kernels (hits through HBM)")
    ap.add_argument("--no-bind", action="store_true", help="keep taxon ids in the table (the fused kernel then gathers {pre,last} by id)")
    ap.add_argument("--rehearse-cpu", action="store_true", help="CPU/gloo rehearsal of the multi-rank plumbing (no GPU, no kernels)")
    ap.add_argument("--rehearse-ipc", action="store_true", help="the whole --gpus N control flow on ONE GPU: N processes share device 0, gloo stands in for "
                                                               "RCCL's process group and the hipIpc communicator (peer-written keys and values) for RCCL's send/recv")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds before self-launched ranks are killed")
    ap.add_argument("--sharded-child", default=None, help=argparse.SUPPRESS)   # internal: run the sharded section in this (child) process
    ap.add_argument("--ipc-child", default=None, help=argparse.SUPPRESS)       # internal: one rank of the two-process leg (file base)
    ap.add_argument("--ipc-name", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--ipc-world", type=int, default=2, help=argparse.SUPPRESS)
    ap.add_argument("--ipc-rank", type=int, default=0, help=argparse.SUPPRESS)
    return ap.parse_args()


def rehearse_cpu(args):
    """The N-rank plumbing of this file without a GPU: rendezvous, ranks_seen, the unique-id hand-off, barriers, max over
    ranks, one JSON line from rank 0.  tests/test_dist_gloo.py runs it at world 2."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(seen)
    box = [bytes(range(128)) if rank == 0 else None]       # stands in for umgap_comm_unique_id()
    dist.broadcast_object_list(box, src=0)
    assert box[0] == bytes(range(128))
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))                     # ranks differ: the max over ranks is rank world-1's time
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * float(t.item()) / max(1, args.steps), 3), "rehearsal": "cpu/gloo: plumbing only, nothing measured",
                          "ranks_seen": int(seen.item())}), flush=True)
    dist.destroy_process_group()


def usable_cores():
    """Threads this process can really run at once: the affinity mask, capped by the cgroup's CPU quota (a GPU box hands out a
    share of the host -- os.cpu_count() still says 256 there)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota != "max" and int(quota) > 0:
                n = min(n, max(1, int(int(quota) / int(period))))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def cpu_baseline(n_tax, ids, parents, valid, rank, taxonomy_handle, d_taxon_of_row, d_row_of_id, device):
    """Oracle ("port") timed on the host cores, bounded sample (DESIGN.md "CPU baseline")."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from umgap_amd import _lib, synth
    from umgap_amd.api import _dev_ptr, _stream_ptr
    cores = usable_cores()
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
    # lookup stage: all cores (rayon in the reference).  The sample is looked up `reps` times inside ONE threaded call, so
    # that starting `cores` threads (tens of ms at 256 threads) does not pass for lookup time
    reps = 8
    qq = np.tile(q, reps)
    t0 = time.perf_counter()
    hits = cpu_index.lookup9(qq, 0, nthreads=cores)[: len(q)]
    t_lookup = (time.perf_counter() - t0) / reps
    del qq
    agg = O.Aggregator(O.Taxonomy.from_arrays(ids, parents, valid, rank), "tree", "lca*")
    offs = np.arange(n_pairs + 1, dtype=np.uint64) * KMERS
    t0 = time.perf_counter()
    agg.taxa2agg_check(hits, offs, lower_bound=2.0, nthreads=1)   # aggregate stage: one thread (taxa2agg)
    t_agg = time.perf_counter() - t0
    # what the reference does NOT do -- the record loop of taxa2agg is serial (src/commands/taxa2agg.rs:161-180) -- but a CPU port
    # could: the aggregate stage over all cores as well (reads are independent); reported beside the faithful figure
    reps_a = 8
    hits_r, offs_r = np.tile(hits, reps_a), np.arange(n_pairs * reps_a + 1, dtype=np.uint64) * KMERS
    t0 = time.perf_counter()
    agg.taxa2agg_check(hits_r, offs_r, lower_bound=2.0, nthreads=cores)
    t_agg_all = (time.perf_counter() - t0) / reps_a
    del hits_r, offs_r
    # the reference runs the stages as concurrent processes: throughput = slowest stage
    value = n_pairs / max(t_lookup, t_agg)
    import shutil
    have_ref = bool(shutil.which("cargo")) or bool(shutil.which("rustc"))
    return {"value": round(value, 1), "unit": "pairs/s", "cores": cores, "kind": "port",
            "value_if_taxa2agg_were_threaded": round(n_pairs / max(t_lookup, t_agg_all), 1),
            "sample": "restated CPU baseline (not the Rust binary: %s on this box), kernel path only (no text): %d pairs (%d lookups) vs a %d-key "
                      "sorted-array index; lookup on %d threads %.3f s per sample (%.1f M lookups/s, %d passes in one threaded call), LCA* -l2 on 1 "
                      "thread %.2f s (as the reference's serial record loop; on all %d threads it would take %.3f s); pipelined stages => rate of the slower one" % (
                          "a Rust toolchain exists but the reference sources do not travel" if have_ref else "no cargo/rustc",
                          n_pairs, n_pairs * KMERS, n_cpu_keys, cores, t_lookup, n_pairs * KMERS / t_lookup / 1e6, reps, t_agg, cores, t_agg_all)}


class stdout_to_stderr:
    """RCCL announces its version on stdout when the first communicator comes up; stdout is for the one JSON line, so fd 1 points at
    stderr meanwhile."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def rehearsing(args):
    return bool(getattr(args, "rehearse_ipc", False))


def rank_device(args, local_rank):
    """The GPU a rank runs on: its own (LOCAL_RANK), or -- rehearsal -- the one GPU every rank shares."""
    return 0 if rehearsing(args) else local_rank


def init_group(args, dist, rank, world, device):
    """RCCL (backend "nccl") between the GPUs of a node; gloo when the ranks rehearse on one GPU (RCCL refuses two ranks on a device).
    Returns the device the bench's own all-reduces use."""
    import torch
    if rehearsing(args):
        dist.init_process_group("gloo", rank=rank, world_size=world)
        return torch.device("cpu")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    return device


def run_sharded_child(args, rank, world, want_first):
    """The sharded section in a CHILD process per rank (own process group on another port, own RCCL communicator): a crash, a GPU
    fault or a stall in the exchange -- the one code path no one-GPU box can exercise beyond a single rank -- then costs the
    "sharded" entry of the line, not the replica measurement that is already complete.  The parent has released its table; the
    answers of batch 0 travel through a file for the bit-for-bit check."""
    import subprocess
    import tempfile

    import torch
    base = os.path.join(tempfile.gettempdir(), "umgap_sharded_%s" % os.environ.get("MASTER_PORT", "0"))
    torch.save((want_first[0].cpu(), want_first[1].cpu()), "%s.rank%d.pt" % (base, rank))
    if rank == 0 and os.path.exists(base):
        os.remove(base)
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29517")) + 17)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)       # the children rendezvous among themselves, not through the launcher's store
    cmd = [sys.executable, os.path.abspath(__file__), "--sharded-child", base, "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--pairs-per-step", str(args.pairs_per_step), "--batches", str(args.batches), "--n-keys", str(args.n_keys), "--n-tax", str(args.n_tax),
           "--load-factor", str(args.load_factor), "--pieces", str(args.pieces)] + (["--rehearse-ipc"] if rehearsing(args) else [])
    tail = ""
    try:
        p = subprocess.run(cmd, env=env, timeout=420, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        status, tail = p.returncode, p.stdout.decode(errors="replace")[-400:]
    except subprocess.TimeoutExpired:
        status = "no answer within 420 s"
    try:
        os.remove("%s.rank%d.pt" % (base, rank))
    except OSError:
        pass
    if rank != 0:
        return None
    if status == 0 and os.path.exists(base):
        res = json.load(open(base))
        os.remove(base)
        res["isolation"] = "child process per rank"
        return res
    return {"error": "the sharded child ended with %s" % (status,), "output_tail": tail}


def measure_traffic_live(args, kernel):
    """HBM bytes per launch of the headline kernel, measured in THIS run: bench.py started twice more as a child under
    `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE: one counter per pass, with --kernel-trace only, as the MI355X guide prescribes;
    KiB -> bytes; FETCH_SIZE counts 64-byte requests), two timed steps each.  The parent has released its table by now.  None when
    rocprofv3 is not there or a pass fails (the committed profiles/traffic.json then stands in, and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not prof:
        return None
    got = {}
    work = tempfile.mkdtemp(prefix="umgap_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_USE_AGENT_STORE")}
    env["TMPDIR"] = "/tmp"
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(work, counter)
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "2", "--warmup", "1", "--batches", "2", "--no-cpu-baseline", "--no-gather", "--no-extras", "--no-sharded", "--pairs-per-step",
                   str(args.pairs_per_step), "--n-keys", str(args.n_keys), "--n-tax", str(args.n_tax), "--load-factor", str(args.load_factor)]
            p = subprocess.run(cmd, env=env, cwd="/tmp", timeout=300, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            if p.returncode != 0:
                return None
            per_dispatch = {}
            for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                with open(path, newline="") as f:
                    for row in csv.DictReader(f):
                        name = row["Kernel_Name"]
                        if (kernel + "<" in name or kernel + "(" in name) and row["Counter_Name"] == counter:
                            per_dispatch[row["Dispatch_Id"]] = per_dispatch.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per_dispatch:
                return None
            got[counter] = sum(per_dispatch.values()) / len(per_dispatch) * 1024.0
            got[counter + "_launches"] = len(per_dispatch)
    except Exception:
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return {"hbm_bytes_per_launch": int(got["FETCH_SIZE"] + got["WRITE_SIZE"]), "fetch_bytes": int(got["FETCH_SIZE"]), "write_bytes": int(got["WRITE_SIZE"]),
            "launches_averaged": min(got["FETCH_SIZE_launches"], got["WRITE_SIZE_launches"])}


def run_ipc_children(args, want_first, local_rank, world=2):
    """The N-rank protocol with every rank its OWN PROCESS on the one GPU of the box (RCCL refuses two ranks on a device, so the
    exchange runs over umgap_comm_init_ipc: the peers' exchange blocks mapped with hipIpc handles, keys and values stored straight into
    them -- UMGAP_SHARD_PUT=0: host-synchronised copies instead).  Rank r holds shard
    r of `world` and classifies the r-th part of the step's pairs; each checks its answers against the replica's.  The ranks
    time-share the GPU, so the figure says what the protocol costs here -- not what N GPUs would do."""
    import subprocess
    import tempfile
    import uuid

    import torch
    base = os.path.join(tempfile.gettempdir(), "umgap_ipc_%d_%s" % (os.getpid(), uuid.uuid4().hex[:8]))
    torch.save((want_first[0].cpu(), want_first[1].cpu()), base + ".want.pt")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_USE_AGENT_STORE")}
    env["UMGAP_SHARD_NO_LOCAL"] = "0"
    name = "/umgap-bench-%s" % uuid.uuid4().hex[:12]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--ipc-child", base, "--ipc-name", name, "--ipc-world", str(world), "--ipc-rank", str(r),
                               "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--pairs-per-step", str(args.pairs_per_step),
                               "--batches", str(args.batches), "--n-keys", str(args.n_keys), "--n-tax", str(args.n_tax), "--load-factor", str(args.load_factor),
                               "--pieces", str(args.pieces)], env=dict(env, UMGAP_BENCH_DEVICE=str(local_rank)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    tails, status = [], []
    for pr in procs:
        try:
            out_b = pr.communicate(timeout=420)[0]
            status.append(pr.returncode)
        except subprocess.TimeoutExpired:
            pr.kill()
            out_b = pr.communicate()[0]
            status.append("no answer within 420 s")
        tails.append(out_b.decode(errors="replace")[-300:])
    res = None
    if all(st == 0 for st in status):
        per = [json.load(open("%s.rank%d.json" % (base, r))) for r in range(world)]
        ms = max(x["ms_per_step"] for x in per)
        how = {"put": "peer-written: the pack stores keys into the owner's receive buffer, the probe stores values into the requester's gather buffer, through "
                      "hipIpc mappings of the exchange blocks; per-piece counters in device memory, no copies, no host barriers",
               "copy": "store and forward: host-synchronised copies out of the peers' send buffers (hipIpc mappings)"}[per[0]["transport"]]
        res = {"ranks": world, "transport": "%d processes on one GPU (umgap_comm_init_ipc), %s" % (world, how), "exchange": per[0]["transport"], "pairs_per_step": sum(x["pairs"] for x in per),
               "ms_per_step": ms, "value": round(sum(x["pairs"] for x in per) / (ms * 1e-3), 1), "unit": "pairs/s", "steps": per[0]["steps"],
               "bit_identical_to_replica": all(x["bit_identical_to_replica"] for x in per), "shard_table_GB_per_rank": per[0]["shard_table_GB"],
               "keys_exchanged_fraction": round(1.0 - 1.0 / world, 3),
               "what": "every rank its own process, shard and reads; the ranks share the GPU, so this is the protocol under the driver's clock, not a scaling figure"}
    else:
        res = {"error": "rank exit status %s" % (status,), "output_tail": tails}
    for f in [base + ".want.pt"] + ["%s.rank%d.json" % (base, r) for r in range(world)]:
        try:
            os.remove(f)
        except OSError:
            pass
    return res


def ipc_child(args):
    """One rank of run_ipc_children."""
    import torch

    from umgap_amd import synth
    from umgap_amd.api import Index
    from umgap_amd.dist import Comm, Sharded
    r, world = args.ipc_rank, args.ipc_world
    dev = int(os.environ.get("UMGAP_BENCH_DEVICE", "0"))
    torch.cuda.set_device(dev)
    device = torch.device("cuda", dev)
    w = synth.SyntheticWorkload(n_tax=args.n_tax, n_keys=args.n_keys, id_space=int(args.n_tax * 1.4), load_factor=args.load_factor, device=dev, build_index=False)
    B = args.pairs_per_step
    lo, hi = B * r // world, B * (r + 1) // world
    n_batches = max(1, min(args.batches, args.steps + args.warmup))
    keys = [w.reads(b * B + lo, hi - lo) for b in range(n_batches)]            # the r-th part of the replica's batches
    want = torch.load(args.ipc_child + ".want.pt")
    want_t, want_s = want[0][lo:hi].to(device), want[1][lo:hi].to(device)
    shard = Index.synthetic(args.n_keys, w.d_taxon_of_row, synth.MUL_A, synth.ADD_B, args.load_factor, r, world)
    comm = Comm.ipc(args.ipc_name, world, r, dev)
    comm.selftest(1 << 20)
    per_piece = -(-(B - B * (world - 1) // world) // args.pieces)
    sh = Sharded(shard, w.taxonomy, comm, per_piece * args.pieces * KMERS, pieces=args.pieces)
    d_taxon = torch.full((hi - lo,), -1, dtype=torch.int32, device=device)
    d_status = torch.full_like(d_taxon, -1)
    steps = max(2, min(5, args.steps))

    def one(i):
        sh.classify9(keys[i % len(keys)], hi - lo, KMERS, d_taxon, d_status, method="tree", strategy="lca*", lower_bound=2.0)

    one(0)
    sh.check()
    torch.cuda.synchronize()
    same = bool(torch.equal(d_taxon, want_t) and torch.equal(d_status, want_s))
    comm.selftest(64)               # collective: everybody starts the timed steps together
    t0 = time.perf_counter()
    for i in range(steps):
        one(i + 1)
    torch.cuda.synchronize()
    comm.selftest(64)               # ... and the clock stops when the last rank is through
    el = time.perf_counter() - t0
    sh.check()
    with open("%s.rank%d.json" % (args.ipc_child, r), "w") as f:
        json.dump({"ms_per_step": round(1e3 * el / steps, 3), "pairs": hi - lo, "steps": steps, "bit_identical_to_replica": same, "transport": sh.transport,
                   "shard_table_GB": round(shard.table_bytes / 1e9, 2)}, f)
    sh.close()
    shard.close()
    comm.close()


def sharded_child(args):
    """Entry of the child started by run_sharded_child: same rank / world / device as its parent (environment), own process group."""
    import torch
    import torch.distributed as dist

    from umgap_amd import synth
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = rank_device(args, int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    init_group(args, dist, rank, world, device)
    w = synth.SyntheticWorkload(n_tax=args.n_tax, n_keys=args.n_keys, id_space=int(args.n_tax * 1.4), load_factor=args.load_factor, device=local_rank,
                                build_index=False)
    B = args.pairs_per_step
    n_batches = max(1, min(args.batches, args.steps + args.warmup))
    keys = [w.reads((rank * n_batches + b) * B, B) for b in range(n_batches)]      # the parent's batches
    want = torch.load("%s.rank%d.pt" % (args.sharded_child, rank))
    want_first = (want[0].to(device), want[1].to(device))
    d_taxon = torch.empty(B, dtype=torch.int32, device=device)
    d_status = torch.empty_like(d_taxon)
    res = sharded_section(args, w, keys, rank, world, device, B, d_taxon, d_status, want_first)
    if rank == 0:
        with open(args.sharded_child + ".tmp", "w") as f:
            json.dump(res, f)
        os.replace(args.sharded_child + ".tmp", args.sharded_child)
    dist.destroy_process_group()


def sharded_section(args, w, keys, rank, world, device, B, d_taxon, d_status, want_first):
    """One sharded step = umgap_sharded_classify9_dev over the rank's B pairs (pack -> RCCL exchange -> probe on the owner ->
    RCCL exchange back -> gather + LCA*), `pieces` pipelined pieces; checked bit for bit against the replica's answers.  Timed as
    shipped (a rank's own 1/N of the keys are probed in place by the consuming kernel) and, at world 1, once more with
    UMGAP_SHARD_NO_LOCAL=1, where EVERY key goes through pack -> exchange -> probe -> gather: what the (N-1)/N of the keys that
    belong to other ranks cost per rank at N > 1."""
    import torch
    import torch.distributed as dist
    from umgap_amd import synth
    from umgap_amd.api import Index
    from umgap_amd.dist import Comm, Sharded
    if w.index is not None:     # the replica has done its job: its HBM goes to the shard and the exchange buffers
        w.index.close()
        w.index = None
    torch.cuda.empty_cache()
    rdev = torch.device("cpu") if rehearsing(args) else device      # where the bench's own all-reduces run (gloo: host tensors)
    seen = torch.ones(1, dtype=torch.int64, device=rdev)
    dist.all_reduce(seen)       # RCCL: every rank of the sharded group is really there
    shard = Index.synthetic(args.n_keys, w.d_taxon_of_row, synth.MUL_A, synth.ADD_B, args.load_factor, rank, world)
    if rehearsing(args):
        comm = Comm.ipc("/umgap-rehearse-%s" % os.environ.get("MASTER_PORT", "0"), world, rank, device.index or 0)
    else:
        comm = Comm.from_torch_distributed(device.index or 0)
    comm.selftest(1 << 20)      # the transport's own send / receive to every rank (itself included), contents checked
    per_piece = -(-B // args.pieces)
    steps = max(2, min(5, args.steps))

    def leg(no_local):
        os.environ["UMGAP_SHARD_NO_LOCAL"] = "1" if no_local else "0"     # read when the group is created
        sh = Sharded(shard, w.taxonomy, comm, per_piece * args.pieces * KMERS, pieces=args.pieces)

        def one(i):
            sh.classify9(keys[i % len(keys)], B, KMERS, d_taxon, d_status, method="tree", strategy="lca*", lower_bound=2.0)

        d_taxon.fill_(-1)
        d_status.fill_(-1)
        one(0)
        sh.check()
        same = bool(torch.equal(d_taxon, want_first[0]) and torch.equal(d_status, want_first[1]))
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            one(i + 1)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=rdev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        sh.check()
        ok = torch.tensor([int(same)], device=rdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        r = {"ms_per_step": round(1e3 * float(el.item()) / steps, 3), "value": round(world * steps * B / float(el.item()), 1), "bit_identical_to_replica": bool(ok.item()),
             "probe_kernel_ms_first_piece": round(sh.probe_ms(), 3), "pieces": sh.pieces, "value_bytes": sh.value_bytes, "wire": sh.wire_bytes_per_step_and_peer(),
             "exchange_mode": sh.transport}
        sh.close()
        return r

    shipped = leg(False)
    out_b, back_b = shipped["wire"]
    res = {"value": shipped["value"], "unit": "pairs/s", "steps": steps, "ms_per_step": shipped["ms_per_step"], "shard_table_GB": round(shard.table_bytes / 1e9, 2),
           "probe_kernel_ms_first_piece": shipped["probe_kernel_ms_first_piece"], "pieces": shipped["pieces"], "value_bytes": shipped["value_bytes"],
           "bit_identical_to_replica": shipped["bit_identical_to_replica"], "ranks_seen": int(seen.item()),
           "transport": ("REHEARSAL on one GPU: %d processes share the device, gloo process group, hipIpc communicator (exchange: %s)" % (world, shipped["exchange_mode"])) if rehearsing(args)
                        else "RCCL (process group of %d)" % world,
           "exchange": "%s inside libumgap_hip.so: 6 B per k-mer out, %d B per hit back; per step and peer %.1f MB out + %.1f MB back "
                       "(%d k-mers per rank and step, %d pieces over 3 streams)" % ("keys and values stored straight into the peers' mapped buffers" if shipped["exchange_mode"] == "put" else "RCCL send/recv groups",
                                                                                   shipped["value_bytes"], out_b / 1e6, back_b / 1e6, B * KMERS, shipped["pieces"])}
    if world == 1:
        every = leg(True)
        res["all_keys_exchanged_ms"] = every["ms_per_step"]
        res["all_keys_exchanged"] = {"ms_per_step": every["ms_per_step"], "value": every["value"], "bit_identical_to_replica": every["bit_identical_to_replica"],
                                     "probe_kernel_ms_first_piece": every["probe_kernel_ms_first_piece"],
                                     "what": "UMGAP_SHARD_NO_LOCAL=1: every key packed, sent (to this rank itself, over RCCL), probed by the owner-side kernel and gathered -- "
                                             "the per-rank compute of the (N-1)/N keys that belong to other ranks at N > 1"}
        res["bit_identical_to_replica"] = bool(res["bit_identical_to_replica"] and every["bit_identical_to_replica"])
        n8 = 8
        res["per_rank_compute_model_ms_at_8_ranks"] = round(shipped["ms_per_step"] / n8 + every["ms_per_step"] * (1 - 1 / n8), 2)
    shard.close()
    comm.close()
    return res


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        # self-launch: one fresh process per rank, started before this process has touched the GPU
        from umgap_amd.dist import spawn_ranks
        rc = spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, timeout=args.launch_timeout)      # (--rehearse-ipc travels in argv)
        if rc:
            sys.stderr.write("bench.py: a rank failed or stalled (exit status %d)\n" % rc)
        sys.exit(rc)
    if args.sharded_child:
        return sharded_child(args)
    if args.ipc_child:
        return ipc_child(args)
    if args.rehearse_cpu:
        return rehearse_cpu(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d; running %d ranks\n" % (args.gpus, world, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    local_rank = rank_device(args, local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    ranks_seen = 1
    rdev = device
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        with stdout_to_stderr():
            rdev = init_group(args, dist, rank, world, device)
            seen = torch.ones(1, dtype=torch.int64, device=rdev)
            dist.all_reduce(seen)                       # RCCL: every rank is really there
            ranks_seen = int(seen.item())
    mode = args.mode
    if mode == "auto":
        mode = "replica"

    from umgap_amd import synth
    from umgap_amd.api import classify9_dev, seedextend_dev
    t_setup = time.perf_counter()
    w = synth.SyntheticWorkload(n_tax=args.n_tax, n_keys=args.n_keys, id_space=int(args.n_tax * 1.4), load_factor=args.load_factor, device=local_rank)
    torch.cuda.synchronize()
    B = args.pairs_per_step
    n_batches = max(1, min(args.batches, args.steps + args.warmup))
    # every rank owns different pairs: rank r, batch b -> pairs [(r*n_batches + b) * B, +B)
    keys = [w.reads((rank * n_batches + b) * B, B) for b in range(n_batches)]
    d_hits = torch.empty(B * KMERS, dtype=torch.int32, device=device) if args.unfused else None
    d_taxon = torch.empty(B, dtype=torch.int32, device=device)
    d_status = torch.empty(B, dtype=torch.int32, device=device)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fused = not args.unfused
    # the answers of batch 0 on the plain (unbound) table: what every later variant must reproduce bit for bit
    classify9_dev(w.index, w.taxonomy, keys[0], B, KMERS, d_taxon, d_status, method="tree", lower_bound=2.0)
    torch.cuda.synchronize()
    want_first = (d_taxon.clone(), d_status.clone())
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
        if timed_idx is not None:
            ev[timed_idx][0].record()
        if fused:
            classify9_dev(w.index, w.taxonomy, k, B, KMERS, d_taxon, d_status, method="tree", lower_bound=2.0,
                          fragments_per_read=2 if args.seedextend else 0, min_seed_size=3, max_gap_size=1)
            if timed_idx is not None:
                ev[timed_idx][1].record()
            return
        w.index.lookup9_dev(k, d_hits, miss_value=0)
        if timed_idx is not None:
            ev[timed_idx][1].record()
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
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity on the last batch: statuses are OK/EMPTY only; and batch 0 through the timed configuration equals the plain one
    st = d_status.cpu().numpy()
    assert ((st == 0) | (st == 1)).all(), "unexpected per-read status in the bench output"
    bit_exact = None
    if not args.seedextend:
        step(0)
        torch.cuda.synchronize()
        bit_exact = bool(torch.equal(d_taxon, want_first[0]) and torch.equal(d_status, want_first[1]))
        assert bit_exact, "the timed kernel's answers differ from the plain table's"

    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)
    lookups = B * KMERS
    # fused: the one kernel carries the whole path (6 724 B per pair); unfused: the probe kernel (76 B per lookup)
    alg_bytes = B * BYTES_PER_PAIR if fused else lookups * BYTES_PER_LOOKUP
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    value = world * args.steps * B / elapsed

    out = {
        "metric": METRIC, "value": round(value, 1), "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 keys / u32 taxa (integer, no MFMA)",
        "data": "synthetic", "ranks_seen": ranks_seen,
        "config": {"workload": "BASELINE.json configs[2]: synthetic 150 bp paired reads (84 packed 9-mers per pair), "
                               "%d resident batches x %d pairs per GPU; lookup%s + taxa2agg -l2 -a lca*%s" % (
                                   n_batches, B, " + seedextend -g1 -s3" if args.seedextend else "", " (one fused kernel)" if fused else ""),
                   "pairs_per_step_per_gpu": B, "n_keys": args.n_keys, "n_tax": args.n_tax,
                   "table_GB": round(w.index.table_bytes / 1e9, 2), "load_factor": args.load_factor,
                   "parallelism": ("1 GPU" if world == 1 else ("REHEARSAL: %d ranks share ONE GPU (the --gpus N control flow; not a scaling figure); " % world if rehearsing(args) else "") +
                                   "replica x%d (table on every GPU, reads split, no data-path collective)" % world),
                   "setup_s": round(t_setup, 1), "table_bound_to_taxonomy": t_bind is not None, "bind_s": None if t_bind is None else round(t_bind, 3),
                   "parity": {"answers_bit_identical_to_unbound_table": bit_exact,
                              "lookup_parity": "contract-only: fst::Map::get is third-party and no reference test touches it (DESIGN.md §2); "
                                               "aggregation pinned by the reference's golden vectors"}},
        "roofline": {"bound": "hbm", "kernel": "k_classify9" if fused else "k_lookup9", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "traffic_source": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(kernel_ms, 3),
                     "frac_on_fused_bytes": round(B * BYTES_PER_PAIR_FUSED / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if fused and kernel_ms > 0 else None,
                     "whole_path_frac": round(value / world * BYTES_PER_PAIR / 1e9 / HBM_PEAK_GBPS, 4)},
    }
    # PMC-derived traffic: NOT measured in this run -- read back from the committed profile of the same kernel and sizes
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            if (tj.get("pairs_per_step") == B and tj.get("n_keys") == args.n_keys and tj.get("kernel") == out["roofline"]["kernel"]
                    and tj.get("load_factor", args.load_factor) == args.load_factor):
                out["roofline"]["traffic"] = tj["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE of %s, tag %s; not re-measured in this run)" % (tj.get("kernel"), tj.get("tag"))
        except Exception:
            pass
    extras = world == 1 and fused and not args.seedextend and not args.no_extras
    if extras:
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
        # sensitivity of the headline to the read mix: the share of k-mers with unrelated
        # taxa decides how many reads need the range-minimum LCA; -l 1 keeps every singleton.  Each leg: one batch generated
        # with that mix, timed on the bound table, and checked bit for bit against the unbound kernel afterwards.
        sweep, sweep_keys = {}, {}
        legs = (("unrelated 10 % (headline mix: 60 % on the root path, 30 % misses)", 60, 10, 2.0), ("unrelated 30 % (40 % path)", 40, 30, 2.0),
                ("unrelated 50 % (20 % path)", 20, 50, 2.0), ("unrelated 10 %, -l 1", 60, 10, 1.0), ("unrelated 50 %, -l 1", 20, 50, 1.0))
        n_s = min(args.steps, 5)
        kbuf = torch.empty(B * KMERS, dtype=torch.int64, device=device)
        got = {}
        for name, pp, up, lb in legs:
            if (pp, up) != sweep_keys.get("mix"):
                w.reads(900_000_000, B, d_keys=kbuf, path_percent=pp, unrelated_percent=up)
                sweep_keys["mix"] = (pp, up)
            classify9_dev(w.index, w.taxonomy, kbuf, B, KMERS, d_taxon, d_status, method="tree", lower_bound=lb)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_s):
                classify9_dev(w.index, w.taxonomy, kbuf, B, KMERS, d_taxon, d_status, method="tree", lower_bound=lb)
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            got[name] = (d_taxon.clone(), d_status.clone())
            sweep[name] = {"value": round(n_s * B / el, 1), "unit": "pairs/s", "ms_per_step": round(1e3 * el / n_s, 3),
                           "classified_below_root": round(float(((d_status == 0) & (d_taxon != 1)).float().mean().item()), 3)}
        out["workload_sweep"] = sweep
    if w.index is not None and w.index.bound:
        w.index.unbind(w.taxonomy)          # what follows wants taxon ids
    if extras:
        # bit-exact check of the sweep legs after the timed region: the unbound kernel on the same batches
        sweep_keys.clear()
        for name, pp, up, lb in legs:
            if (pp, up) != sweep_keys.get("mix"):
                w.reads(900_000_000, B, d_keys=kbuf, path_percent=pp, unrelated_percent=up)
                sweep_keys["mix"] = (pp, up)
            classify9_dev(w.index, w.taxonomy, kbuf, B, KMERS, d_taxon, d_status, method="tree", lower_bound=lb)
            torch.cuda.synchronize()
            sweep[name]["bit_identical_to_unbound_kernel"] = bool(torch.equal(d_taxon, got[name][0]) and torch.equal(d_status, got[name][1]))
        del kbuf, got
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
    sharded_failed = False
    # one GPU, default run: the sharded step still runs -- in a child with a process group of one over RCCL -- so that the line carries
    # driver-timed evidence of the exchange path (as shipped and with every key exchanged)
    solo_sharded = not multi and world == 1 and args.mode != "replica" and fused and not args.seedextend and not args.no_sharded and not args.no_extras
    if solo_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            from umgap_amd.dist import free_port
            os.environ["MASTER_PORT"] = str(free_port())
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local_rank))
    if (multi or solo_sharded) and args.mode != "replica":
        # the replica has done its job: its HBM goes to the child's shard and exchange buffers
        if w.index is not None:
            w.index.close()
            w.index = None
        keys = None
        torch.cuda.empty_cache()
        res = run_sharded_child(args, rank, world, want_first)
        if rank == 0:
            out["sharded"] = res
            sharded_failed = "value" not in res
        if solo_sharded and not sharded_failed:
            out["sharded"]["two_processes_one_gpu"] = run_ipc_children(args, want_first, local_rank)
        if rank == 0 and args.mode == "sharded" and "value" in out["sharded"]:
            out["replica"] = {"value": out["value"], "ms_per_step": out["ms_per_step"]}
            out["value"], out["ms_per_step"] = out["sharded"]["value"], out["sharded"]["ms_per_step"]
            out["config"]["parallelism"] = "sharded x%d (bucket-range shards, RCCL exchange)" % world
    if rank == 0 and world == 1 and extras and fused and not getattr(args, "no_live_traffic", False):
        # roofline.traffic measured in this very run (two short passes of this script under rocprofv3 --pmc; the table of this process is
        # released by now, or is released here)
        if w.index is not None:
            w.index.close()
            w.index = None
        keys = None
        torch.cuda.empty_cache()
        live = measure_traffic_live(args, out["roofline"]["kernel"])
        if live:
            out["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (one pass each) over %d launches of %s "
                                                 "in child runs of this script (fetch %d B + write %d B per launch)" % (
                                                     live["launches_averaged"], out["roofline"]["kernel"], live["fetch_bytes"], live["write_bytes"]))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.n_tax, w.ids, w.parents, w.valid, w.rank, w.taxonomy._h, w.d_taxon_of_row, w.d_row_of_id, device)
    if rank == 0 and world == 1 and extras and not args.no_end_to_end and os.path.exists(os.path.join(ROOT, "umgap_amd", "bin", "umgap")):
        # text in -> text out through the unchanged script's chain, beside the oracle's same four stages (tools/end_to_end.py)
        if w.index is not None:
            w.index.close()
            w.index = None
        keys = None
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        try:
            # BASELINE.json configs[4] as one device call per batch: the tryptic presets of umgap-analyse.sh:289-298 (digest -> length
            # window -> variable-length lookup -> MRTL) over a 1 G-key tryptic index (SURVEY.md §8d Config 5; tools/bench_tryptic_fused.py);
            # outside the timed calls 20 k sampled pairs per preset go through the checker (regex digest, closed-form membership, the
            # oracle's literal aggregator)
            import bench_tryptic_fused
            out["tryptic_pipeline"] = bench_tryptic_fused.run(1_000_000_000, 10_000_000, steps=3, workload=w)
        except Exception as e:  # noqa: BLE001
            out["tryptic_pipeline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
        try:
            # surface B (protein fragments in) over the plain table and over the same proteome's index in walk order (index_walk.hip): reads
            # that are proteome fragments with 3 % of their residues changed; the walk answers checked bit for bit (tools/bench_surface_b.py)
            import bench_surface_b
            legs = bench_surface_b.ALL_LEGS[:2]
            plain = bench_surface_b.run(50_000_000, 10_000_000, 30, walk=False, steps=3, workload=w, legs=legs)
            torch.cuda.empty_cache()
            walk = bench_surface_b.run(50_000_000, 10_000_000, 30, walk=True, steps=3, workload=w, legs=legs)
            out["surface_b"] = {"index_keys": plain["index_keys"], "reads": "10 M pairs of 2 x 50 residues, proteome fragments with 3 % of the residues changed",
                                "plain_table": {k: plain[k] for k, *_ in legs}, "walk_order_index": {k: walk[k] for k, *_ in legs},
                                "walk_tables_GB": [walk["walk_table_GB"], walk["walk_array_GB"]], "plain_table_GB": plain["table_GB"]}
        except Exception as e:  # noqa: BLE001
            out["surface_b"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
        try:
            # surface A with OVERLAPPING k-mers (what pept2lca really gets: every window of a fragment, prot2kmer.rs) over the plain table and
            # over the walk-order index of the same proteome through umgap_classify9_dev: one probe per run of neighbouring 9-mers
            # (tools/bench_surface_a_walk.py; requests per k-mer from the PMC pass in profiles/r05_surface_a_pmc.json)
            import bench_surface_a_walk
            sa = bench_surface_a_walk.run(50_000_000, 10_000_000, 30, steps=3, workload=w)
            try:
                prof = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_surface_a_pmc.json")))["runs"]
                sa["requests_per_kmer"] = {"source": "profiles/r05_surface_a_pmc.json (TCC_EA0_RDREQ_sum per k-mer, key stream of 0.125 included; not re-measured in this run)",
                                           "plain_table": prof["3.0 % of the residues replaced"]["requests_per_kmer_plain_table"],
                                           "walk_order_index": prof["3.0 % of the residues replaced"]["requests_per_kmer_walk_index_runs_cut_at_fragments"],
                                           "walk_order_index_reads_that_follow_the_proteome": prof["0.0 % of the residues replaced"]["requests_per_kmer_walk_index_runs_cut_at_fragments"]}
            except (OSError, KeyError, ValueError):
                pass
            out["surface_a_overlapping"] = sa
            torch.cuda.empty_cache()
            # the same with reads that follow the proteome exactly (0 % of the residues replaced): what the walk order is for
            s0 = bench_surface_a_walk.run(50_000_000, 10_000_000, 0, steps=2, workload=w)
            sa["reads_that_follow_the_proteome"] = {k: s0[k] for k in ("plain_table", "walk_order_index", "speedup_walk_over_plain")}
        except Exception as e:  # noqa: BLE001
            out["surface_a_overlapping"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
        try:
            # a pair translated in all six frames (12 fragments, 504 windows: the sensitivity pipelines, umgap-analyse.sh:277-288) through the
            # LONG variant of the fused kernel, beside the 2 x 50 gene-fragment form of the same pairs (tools/bench_six_frame.py)
            import bench_six_frame
            out["six_frame"] = bench_six_frame.run(50_000_000, 2_000_000, 30, steps=3, workload=w)
        except Exception as e:  # noqa: BLE001
            out["six_frame"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
        try:
            # the host-pointer entry points a Rust host binds, the PCIe link included (never `value`): pinned arrays in, pinned arrays out
            import bench_host_lookup
            import bench_records
            hl = bench_host_lookup.run(2_000_000_000, 200_000_000, steps=3)
            torch.cuda.empty_cache()
            hr = bench_records.run(50_000_000, 10_000_000, steps=3)
            out["host_fed"] = {"umgap_index_lookup9": {"G_lookups_per_s": hl["pinned"]["G_lookups_per_s"], "GB_per_s_in": hl["pinned"]["GB_per_s_in"],
                                                       "G_lookups_per_s_min_median_max": hl["pinned"]["G_lookups_per_s_min_median_max"],
                                                       "pageable_G_lookups_per_s": hl["pageable"]["G_lookups_per_s"], "queries": hl["queries"], "calls": hl["pinned"]["calls"]},
                               "umgap_classify_records": {"pairs_per_s": hr["pinned"]["pairs_per_s"], "GB_per_s_in": hr["pinned"]["GB_per_s_in"],
                                                          "M_pairs_per_s_min_median_max": hr["pinned"]["M_pairs_per_s_min_median_max"], "route": hr["pinned"]["route"],
                                                          "pageable_pairs_per_s": hr["pageable"]["pairs_per_s"], "pairs": hr["pairs"], "pipeline": hr["pipeline"], "calls": hr["pinned"]["calls"]},
                               "link_GBps_bare_copy": hr["h2d_pinned_32MB_pieces_GBps"]}
        except Exception as e:  # noqa: BLE001
            out["host_fed"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
        try:
            import end_to_end
            out["end_to_end"], out["cpu_baseline_end_to_end"] = end_to_end.run(8_000_000, 200_000)
        except Exception as e:  # noqa: BLE001
            out["end_to_end"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()
    if sharded_failed and args.mode == "sharded":
        sys.exit(3)      # the sharded step WAS the headline: no number, no success


if __name__ == "__main__":
    main()
