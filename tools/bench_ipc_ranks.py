#!/usr/bin/env python3
"""The sharded step with every rank its own PROCESS on the one GPU of the box (umgap_comm_init_ipc), timed: what bench.py's
`sharded.two_processes_one_gpu` leg measures, without the 128 GB replica in front of it (every rank checks its answers against a whole
table of its own first, then frees it).  The environment goes through to the ranks: UMGAP_SHARD_PUT=0 (copies instead of peer-written
keys and values), UMGAP_PUT_FENCE, UMGAP_SHARD_NO_LOCAL=1 ...
    python tools/bench_ipc_ranks.py [world 2] [pairs per step, all ranks together 10000000] [n_keys 2000000000] [pieces 8] [steps 5]"""
import json
import os
import subprocess
import sys
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KMERS = 84


def child(name, world, r, B, n_keys, pieces, steps, out_path):
    import faulthandler
    faulthandler.enable()
    if os.environ.get("IPC_BENCH_GDB"):
        import ctypes
        ctypes.CDLL(None).prctl(0x59616d61, ctypes.c_ulong(-1 & 0xFFFFFFFFFFFFFFFF), 0, 0, 0)     # PR_SET_PTRACER, PR_SET_PTRACER_ANY
    import torch
    from umgap_amd import api, synth
    from umgap_amd.dist import Comm, Sharded
    torch.cuda.set_device(0)
    w = synth.SyntheticWorkload(n_tax=2_500_000, n_keys=n_keys, id_space=3_500_000, load_factor=0.5)
    lo, hi = B * r // world, B * (r + 1) // world
    keys = [w.reads(b * B + lo, hi - lo) for b in range(3)]
    want_t = torch.empty(hi - lo, dtype=torch.int32, device="cuda")
    want_s = torch.empty_like(want_t)
    api.classify9_dev(w.index, w.taxonomy, keys[0], hi - lo, KMERS, want_t, want_s, method="tree", lower_bound=2.0)
    torch.cuda.synchronize()
    w.index.close()
    w.index = None
    torch.cuda.empty_cache()
    shard = api.Index.synthetic(n_keys, w.d_taxon_of_row, synth.MUL_A, synth.ADD_B, 0.5, r, world)
    comm = Comm.ipc(name, world, r, 0)
    per_piece = -(-(B - B * (world - 1) // world) // pieces)
    sh = Sharded(shard, w.taxonomy, comm, per_piece * pieces * KMERS, pieces=pieces)
    d_t = torch.full((hi - lo,), -1, dtype=torch.int32, device="cuda")
    d_s = torch.full_like(d_t, -1)

    def one(i):
        sh.classify9(keys[i % len(keys)], hi - lo, KMERS, d_t, d_s, method="tree", strategy="lca*", lower_bound=2.0)

    one(0)
    sh.check()
    torch.cuda.synchronize()
    same = bool(torch.equal(d_t, want_t) and torch.equal(d_s, want_s))
    one(1)
    torch.cuda.synchronize()
    comm.selftest(64)               # collective: everybody starts the timed steps together
    t0 = time.perf_counter()
    for i in range(steps):
        one(i + 2)
    torch.cuda.synchronize()
    comm.selftest(64)               # ... and the clock stops when the last rank is through
    el = time.perf_counter() - t0
    sh.check()
    with open(out_path, "w") as f:
        json.dump({"ms_per_step": round(1e3 * el / steps, 3), "pairs": hi - lo, "bit_identical_to_whole_table": same, "exchange": sh.transport}, f)
    sh.close()
    shard.close()
    comm.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        a = sys.argv[2:]
        return child(a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), a[7])
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    n_keys = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000_000
    pieces = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    name = "/umgap-ipcbench-%s" % uuid.uuid4().hex[:12]
    base = "/tmp/umgap_ipcbench_%s" % uuid.uuid4().hex[:8]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", name, str(world), str(r), str(B), str(n_keys), str(pieces), str(steps),
                               "%s.%d.json" % (base, r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    tails = []
    for p in procs:
        try:
            tails.append(p.communicate(timeout=float(os.environ.get("IPC_BENCH_TIMEOUT", "300")))[0].decode(errors="replace")[-1500:])
        except subprocess.TimeoutExpired:
            import signal
            where = []
            for tid in sorted(os.listdir("/proc/%d/task" % p.pid))[:40]:
                try:
                    st = open("/proc/%d/task/%s/stat" % (p.pid, tid)).read().split()
                    where.append("%s %s %s wchan=%s syscall=%s" % (tid, st[1], st[2], open("/proc/%d/task/%s/wchan" % (p.pid, tid)).read().strip(),
                                                                 open("/proc/%d/task/%s/syscall" % (p.pid, tid)).read().strip()[:60]))
                except OSError as e:
                    where.append("%s: %s" % (tid, e))
            print("rank pid %d threads:\n  " % p.pid + "\n  ".join(w_ for k, w_ in enumerate(where) if k == 0 or "futex_do_wait" not in w_), file=sys.stderr, flush=True)
            if os.environ.get("IPC_BENCH_GDB") and os.path.exists("/opt/rocm/bin/rocgdb"):
                g = subprocess.run(["/opt/rocm/bin/rocgdb", "-p", str(p.pid), "-batch", "-ex", "thread 1", "-ex", "bt 40"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
                print("rocgdb:\n" + g.stdout.decode(errors="replace")[-6000:], file=sys.stderr, flush=True)
            p.send_signal(signal.SIGABRT)           # faulthandler prints where the rank stood
            try:
                tails.append("no answer in time: " + p.communicate(timeout=10)[0].decode(errors="replace")[-3000:])
            except subprocess.TimeoutExpired:
                p.kill()
                tails.append("no answer in time, and none to SIGABRT: " + p.communicate()[0].decode(errors="replace")[-1500:])
    if any(p.returncode != 0 for p in procs):
        print(json.dumps({"error": [p.returncode for p in procs], "tails": tails}))
        return 1
    per = [json.load(open("%s.%d.json" % (base, r))) for r in range(world)]
    for r in range(world):
        os.remove("%s.%d.json" % (base, r))
    ms = max(x["ms_per_step"] for x in per)
    print(json.dumps({"ranks": world, "exchange": per[0]["exchange"], "pairs_per_step": B, "n_keys": n_keys, "pieces": pieces, "steps": steps, "ms_per_step": ms,
                      "ms_per_rank": [x["ms_per_step"] for x in per], "M_pairs_per_s": round(B / ms / 1e3, 1),
                      "bit_identical_to_whole_table": all(x["bit_identical_to_whole_table"] for x in per),
                      "env": {k: v for k, v in os.environ.items() if k.startswith("UMGAP_")}}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
