"""Sharded multi-GPU lookup (BASELINE.json north_star): one process per GPU, the
index sharded by bucket range, read keys / hits exchanged with torch.distributed all-to-all
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Per step on every rank (B local pairs, K = 84 B keys):
    partition : keys -> N regions of `cap` slots by owner shard (HIP kernel: LDS counters per 4096-key tile,
                one global atomic per owner and tile)
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

    # two side streams: chunk c+1's partition / probe run while chunk c's all-to-all is on the wire
    def streams(self, n):
        return [self.torch.cuda.Stream(device=self.device) for _ in range(n)]

    def stream_ctx(self, s):
        return self.torch.cuda.stream(s)

    def fork(self, side):
        side.wait_stream(self.torch.cuda.current_stream())

    def join(self, side):
        self.torch.cuda.current_stream().wait_stream(side)


class ShardedLookup:
    """lookup(keys) -> hits across `world` ranks, each owning one shard of the table."""

    def __init__(self, ops, rank, world, max_keys, slack=0.02, chunks=1):
        """chunks > 1 (GPU only): the batch is cut into that many pieces that alternate over two side
        streams, so the exchange of one piece overlaps the partition / probe / gather of the next."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ops, self.rank, self.world = ops, rank, world
        self.chunks = max(1, int(chunks)) if hasattr(ops, "streams") else 1
        self.max_keys = max_keys
        per = -(-max_keys // self.chunks)
        self.cap = int(per / world * (1.0 + slack)) + 4096
        n = self.cap * world
        nbuf = 2 if self.chunks > 1 else 1
        self.buf = [dict(send_keys=ops.empty(n, torch.int64), recv_keys=ops.empty(n, torch.int64), recv_taxa=ops.empty(n, torch.int32),
                         back_taxa=ops.empty(n, torch.int32), perm=ops.empty(per, torch.int32), cursor=ops.empty(world, torch.int64),
                         overflow=ops.empty(1, torch.int32)) for _ in range(nbuf)]
        self.side = ops.streams(2) if self.chunks > 1 else None

    def _one(self, b, keys, hits, events=None):
        perm = b["perm"][: keys.numel()]
        self.ops.partition(keys, self.world, self.cap, b["send_keys"], perm, b["cursor"], b["overflow"])
        self.dist.all_to_all_single(b["recv_keys"], b["send_keys"])
        if events is not None:
            events[0].record()
        self.ops.lookup(b["recv_keys"], b["recv_taxa"])
        if events is not None:
            events[1].record()
        self.dist.all_to_all_single(b["back_taxa"], b["recv_taxa"])
        self.torch.index_select(b["back_taxa"], 0, perm, out=hits)

    def lookup(self, keys, hits, events=None):
        """keys int64[K] -> hits int32[K] (0 = miss).  `events` = (start, end) timing events around
        the local probe kernel of the (first) piece, recorded on the stream it runs on."""
        K = keys.numel()
        if self.chunks == 1:
            self._one(self.buf[0], keys, hits, events)
            return hits
        per = -(-K // self.chunks)
        for s in self.side:
            self.ops.fork(s)
        for c in range(self.chunks):
            lo, hi = c * per, min(K, (c + 1) * per)
            if lo >= hi:
                break
            with self.ops.stream_ctx(self.side[c & 1]):
                # pieces c and c+2 share a stream and a buffer set: stream order keeps them apart
                self._one(self.buf[c & 1], keys[lo:hi], hits[lo:hi], events if c == 0 else None)
        for s in self.side:
            self.ops.join(s)
        return hits

    def check_overflow(self):
        if any(int(b["overflow"].item()) != 0 for b in self.buf):
            raise RuntimeError("sharded lookup: a send region overflowed its capacity; raise `slack`")
