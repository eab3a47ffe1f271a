"""Sharded multi-GPU lookup (BASELINE.json north_star): one process per GPU, the
index sharded by bucket range, read keys / hits exchanged with torch.distributed all-to-all
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Per step on every rank (B local pairs, K = 84 B keys):
    partition : keys -> N regions of `cap` slots by owner shard (HIP kernel, one atomic per wave and owner)
    exchange  : all_to_all_single(keys)          8 B per key, each rank sends ~K/N to every peer
    probe     : local table shard (the same quad-probe kernel)
    exchange  : all_to_all_single(taxa)          4 B per key back
    gather    : hits[i] = returned[perm[i]]
Regions have a fixed capacity (keys are spread by a bijective mixer, so every region receives
K/N +- a few sigma); unused slots carry an invalid key that the probe answers with a miss, which
keeps both collectives equal-split and free of any host synchronisation.  The reads themselves are
never exchanged: each rank aggregates the reads it owns (no data-path collective after the hits).
"""
import ctypes as C


class GpuOps:
    """The device side of one rank: HIP kernels through the C ABI, buffers as torch CUDA tensors."""

    def __init__(self, index, device):
        import torch
        self.torch = torch
        self.index = index
        self.device = device

    def empty(self, n, dtype):
        return self.torch.empty(n, dtype=dtype, device=self.device)

    def partition(self, keys, n_shards, cap, send, perm, cursor, overflow):
        from . import _lib
        from .api import _dev_ptr, _stream_ptr
        send.fill_(-1)      # invalid key (bits above 45 set): answered with a miss
        cursor.zero_()
        overflow.zero_()
        _lib.check(_lib.load().umgap_index_partition9_dev(self.index._h, _dev_ptr(keys), keys.numel(), n_shards, cap, _dev_ptr(send),
                                                          _dev_ptr(perm), _dev_ptr(cursor), _dev_ptr(overflow), _stream_ptr(None)))

    def lookup(self, keys, out):
        self.index.lookup9_dev(keys, out, miss_value=0)


class ShardedLookup:
    """lookup(keys) -> hits across `world` ranks, each owning one shard of the table."""

    def __init__(self, ops, rank, world, max_keys, slack=0.02):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ops, self.rank, self.world = ops, rank, world
        self.cap = int(max_keys / world * (1.0 + slack)) + 4096
        n = self.cap * world
        self.send_keys = ops.empty(n, torch.int64)
        self.recv_keys = ops.empty(n, torch.int64)
        self.recv_taxa = ops.empty(n, torch.int32)
        self.back_taxa = ops.empty(n, torch.int32)
        self.perm = ops.empty(max_keys, torch.int64)
        self.cursor = ops.empty(world, torch.int64)
        self.overflow = ops.empty(1, torch.int32)

    def lookup(self, keys, hits, events=None):
        """keys int64[K] -> hits int32[K] (0 = miss).  `events` = (start, end) timing events around
        the local probe kernel, recorded on the current stream."""
        perm = self.perm[: keys.numel()]
        self.ops.partition(keys, self.world, self.cap, self.send_keys, perm, self.cursor, self.overflow)
        self.dist.all_to_all_single(self.recv_keys, self.send_keys)
        if events is not None:
            events[0].record()
        self.ops.lookup(self.recv_keys, self.recv_taxa)
        if events is not None:
            events[1].record()
        self.dist.all_to_all_single(self.back_taxa, self.recv_taxa)
        self.torch.index_select(self.back_taxa, 0, perm, out=hits)
        return hits

    def check_overflow(self):
        if int(self.overflow.item()) != 0:
            raise RuntimeError("sharded lookup: a send region overflowed its capacity; raise `slack`")
