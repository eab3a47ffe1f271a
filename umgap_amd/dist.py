"""Multi-GPU harness over the C ABI's sharded index (include/umgap_hip.h "the index sharded over the GPUs of one
node"; SURVEY.md §8e, BASELINE.json north_star).  The exchange itself -- RCCL send/recv groups over xGMI, 6 bytes per
k-mer out and 3 per hit back, pipelined pieces -- lives in libumgap_hip.so (umgap_amd/csrc/shard.hip); this module only
creates the communicator and wraps the handles.  torch.distributed is used for what the bench contract asks of it
(rendezvous, barrier, max-over-ranks of the elapsed time) and to hand rank 0's RCCL unique id to the other ranks.

    Comm.from_torch_distributed()      one rank per process, RCCL (GPU box)
    Comm.local(world)                  all ranks in this process on one GPU: rehearsal of the N-rank protocol
    Sharded(shard_index, taxonomy, comm, max_keys, pieces)   .lookup9 / .classify9 / .check
"""
import ctypes as C
import os
import subprocess
import sys

from . import _lib
from ._lib import check

UNIQUE_ID_BYTES = 128


class Comm:
    def __init__(self, handle, transport):
        self._h = C.c_void_p(handle)
        self.transport = transport

    @property
    def world(self):
        return _lib.load().umgap_comm_world(self._h)

    @property
    def rank(self):
        return _lib.load().umgap_comm_rank(self._h)

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
        check(_lib.load().umgap_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def init_rank(cls, unique_id, world, rank, device):
        assert len(unique_id) == UNIQUE_ID_BYTES
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        check(_lib.load().umgap_comm_init_rank(buf, world, rank, device, C.byref(h)))
        return cls(h.value, "rccl")

    @classmethod
    def from_torch_distributed(cls, device):
        """Collective over the default process group: rank 0 draws the RCCL unique id, everybody gets it."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls.init_rank(box[0], world, rank, device)

    @classmethod
    def local(cls, world, device=0):
        """All `world` ranks at once (one GPU, device-to-device exchange); drive each from its own thread."""
        arr = (C.c_void_p * world)()
        check(_lib.load().umgap_comm_init_local(world, device, arr))
        return [cls(arr[r], "local") for r in range(world)]

    @classmethod
    def ipc(cls, name, world, rank, device=0):
        """Collective over `world` PROCESSES (one GPU each, or sharing one): hipIpc memory handles, no RCCL.  `name` = "/something"."""
        h = C.c_void_p()
        check(_lib.load().umgap_comm_init_ipc(name.encode(), world, rank, device, C.byref(h)))
        return cls(h.value, "ipc")

    def selftest(self, bytes_per_peer=1 << 20):
        """Collective: a known pattern through the lookup's own transport, to every rank including this one."""
        check(_lib.load().umgap_comm_selftest(self._h, bytes_per_peer))

    def close(self):
        if self._h:
            _lib.load().umgap_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Sharded:
    """One rank's view of the table sharded over `comm.world` GPUs."""

    def __init__(self, shard_index, taxonomy, comm, max_keys, pieces=8, slack=0.03):
        self.index, self.taxonomy, self.comm = shard_index, taxonomy, comm   # borrowed by the C handle: keep them alive
        h = C.c_void_p()
        check(_lib.load().umgap_sharded_create(shard_index._h, taxonomy._h if taxonomy is not None else None, comm._h, max_keys, pieces, slack,
                                               C.byref(h)))
        self._h = h

    @property
    def value_bytes(self):
        return _lib.load().umgap_sharded_value_bytes(self._h)

    @property
    def region_slots(self):
        return _lib.load().umgap_sharded_region_slots(self._h)

    @property
    def pieces(self):
        return _lib.load().umgap_sharded_pieces(self._h)

    @property
    def transport(self):
        """"put": keys and values are stored straight into the peers' buffers (hipIpc mappings, device-side counters);
        "copy": store and forward through the communicator (RCCL send/recv groups, or copies)."""
        return "put" if _lib.load().umgap_sharded_transport(self._h) else "copy"

    def wire_bytes_per_step_and_peer(self):
        """(out, back): bytes this rank sends to EACH peer per call (keys) and gets back from it (values)."""
        return 6 * self.region_slots * self.pieces, self.value_bytes * self.region_slots * self.pieces

    def lookup9(self, d_keys45, d_out, miss_value=0, stream=None):
        from .api import _dev_ptr, _stream_ptr
        check(_lib.load().umgap_sharded_lookup9_dev(self._h, _dev_ptr(d_keys45), d_keys45.numel(), miss_value, _dev_ptr(d_out), _stream_ptr(stream)))
        return d_out

    def classify9(self, d_keys45, n_reads, kmers_per_read, d_out_taxon, d_out_status, method="tree", strategy="lca*", factor=0.25,
                  lower_bound=0.0, ranked_only=False, fragments_per_read=0, min_seed_size=2, max_gap_size=0, stream=None):
        from .api import METHODS, STRATEGIES, _dev_ptr, _stream_ptr
        check(_lib.load().umgap_sharded_classify9_dev(self._h, METHODS[method], STRATEGIES[strategy], factor, lower_bound, int(ranked_only),
                                                      _dev_ptr(d_keys45), n_reads, kmers_per_read, fragments_per_read, min_seed_size, max_gap_size,
                                                      _dev_ptr(d_out_taxon), _dev_ptr(d_out_status), _stream_ptr(stream)))

    def check(self):
        """Synchronises; raises if an exchange region overflowed since the last check."""
        check(_lib.load().umgap_sharded_check(self._h))

    def probe_ms(self):
        ms = C.c_float()
        check(_lib.load().umgap_sharded_probe_ms(self._h, C.byref(ms)))
        return ms.value

    def close(self):
        if self._h:
            _lib.load().umgap_sharded_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(script, argv, world, timeout=None, extra_env=None):
    """Starts `world` copies of `script argv` as one process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment, rendezvous on 127.0.0.1) and waits for them: what `python bench.py --gpus N` does when it was not started
    under torch.distributed.run.  Called BEFORE the parent touches the GPU (children are fresh processes, the parent never
    re-executes itself).  Returns the largest exit code; a rank that outlives `timeout` seconds is killed (exit code 124)."""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "UMGAP_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    rc = 0
    import time
    deadline = None if timeout is None else time.time() + timeout
    for p in procs:
        try:
            p.wait(None if deadline is None else max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            rc = max(rc, 124)
    for p in procs:
        if p.poll() is None:
            p.kill()
            p.wait()
        rc = max(rc, abs(p.returncode))
    return rc
