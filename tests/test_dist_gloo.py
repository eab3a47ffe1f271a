"""The N > 1 path on CPU: two gloo ranks run the sharded-lookup exchange (partition -> all-to-all
keys -> local probe -> all-to-all taxa -> gather) with numpy stand-ins for the two HIP kernels.
What is under test is the distributed plumbing of umgap_amd/dist.py (routing, fixed-capacity
regions, padding, permutation), which is identical on the GPU box."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from umgap_amd.dist import ShardedLookup


class CpuOps:
    """Owner = key mod world; each rank answers only the keys it owns from a dict."""

    def __init__(self, rank, world, table):
        self.rank, self.world = rank, world
        self.table = {k: v for k, v in table.items() if k % world == rank}

    def empty(self, n, dtype):
        return torch.empty(n, dtype=dtype)

    def partition(self, keys, n_shards, cap, send, perm, cursor, overflow):
        send.fill_(-1)
        cursor.zero_()
        overflow.zero_()
        k = keys.numpy()
        for i, key in enumerate(k):
            o = int(key) % n_shards
            pos = int(cursor[o])
            if pos < cap:
                send[o * cap + pos] = int(key)
                perm[i] = o * cap + pos
            else:
                perm[i] = 0
                overflow[0] = 1
            cursor[o] += 1

    def lookup(self, keys, out):
        k = keys.numpy()
        out.copy_(torch.tensor([self.table.get(int(x), 0) if x >= 0 else 0 for x in k], dtype=torch.int32))


def _worker(rank, world, port, n_keys, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1234)
    universe = rng.choice(1 << 40, 5000, replace=False)
    table = {int(k): int(v) for k, v in zip(universe, rng.integers(1, 1 << 30, len(universe)))}
    rng_r = np.random.default_rng(100 + rank)
    # each rank owns different reads: a mix of members and strangers, ragged vs the region size
    mine = np.concatenate([rng_r.choice(universe, n_keys - n_keys // 3), rng_r.integers(1 << 41, 1 << 42, n_keys // 3)])
    rng_r.shuffle(mine)
    sl = ShardedLookup(CpuOps(rank, world, table), rank, world, n_keys, slack=0.5)
    hits = torch.empty(n_keys, dtype=torch.int32)
    for _ in range(2):  # buffers are reused across steps
        sl.lookup(torch.from_numpy(mine.astype(np.int64)), hits)
    sl.check_overflow()
    want = np.array([table.get(int(k), 0) for k in mine], np.int32)
    ok = bool((hits.numpy() == want).all()) and bool((want != 0).any())
    # a skewed batch must trip the overflow check instead of silently dropping keys
    skew = torch.from_numpy(np.full(n_keys, int(universe[0]) // world * world, np.int64))
    tight = ShardedLookup(CpuOps(rank, world, table), rank, world, n_keys, slack=0.0)
    tight.cap = n_keys // world   # regions too small for a batch that goes to one owner
    tight.lookup(skew, hits)
    try:
        tight.check_overflow()
        tripped = False
    except RuntimeError:
        tripped = True
    flag = torch.tensor([int(ok), int(tripped)])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        results.put(flag.tolist())
    dist.destroy_process_group()


def test_sharded_lookup_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    results = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 3001, results)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, tripped = results.get(timeout=10)
    assert ok == 1 and tripped == 1
