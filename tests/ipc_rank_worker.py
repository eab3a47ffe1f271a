"""One rank of the process-per-rank sharded test (tests/test_gpu_parity.py::test_sharded_table_ranks_as_processes): its own
process, its own HIP context, its own shard of the table; the exchange goes through umgap_comm_init_ipc.  Every rank builds
the same synthetic workload, answers its own reads from the whole table first and then through the sharded path, and exits 0
only if the two agree bit for bit.  argv: shm name, world, rank, pieces, transport ("put" / "copy": what the group must report)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    name, world, rank, pieces = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import torch
    from umgap_amd import api
    from umgap_amd.dist import Comm, Sharded
    from umgap_amd.synth import ADD_B, MUL_A, SyntheticWorkload
    torch.cuda.set_device(0)
    n_tax, n_keys, stride = 20_000, 1_500_000, 84
    w = SyntheticWorkload(n_tax=n_tax, n_keys=n_keys)
    n_reads = 3001 + 173 * rank                                # ragged: every rank owns a different number of reads
    max_keys = (3001 + 173 * (world - 1) + pieces) // pieces * pieces * stride
    keys = w.reads(10_000 * rank, n_reads, kmers_per_pair=stride)
    want_hits = w.index.lookup9_dev(keys, miss_value=0).clone()
    modes = [("tree", "lca*", 0, 2.0), ("tree", "hybrid", 2, 1.0), ("rmq", "mrtl", 2, 1.0)]
    want = {}
    for m in modes:
        t = torch.empty(n_reads, dtype=torch.int32, device="cuda")
        s = torch.empty_like(t)
        api.classify9_dev(w.index, w.taxonomy, keys, n_reads, stride, t, s, method=m[0], strategy=m[1], factor=0.25, lower_bound=m[3],
                          fragments_per_read=m[2], min_seed_size=3, max_gap_size=1)
        want[m] = (t, s)
    torch.cuda.synchronize()
    shard = api.Index.synthetic(n_keys, w.d_taxon_of_row, MUL_A, ADD_B, 0.0, rank, world)
    comm = Comm.ipc(name, world, rank, 0)
    assert comm.world == world and comm.rank == rank and comm.transport == "ipc"
    comm.selftest(3 << 16)
    sh = Sharded(shard, w.taxonomy, comm, max_keys, pieces=pieces)
    if len(sys.argv) > 5:
        assert sh.transport == sys.argv[5], (sh.transport, sys.argv[5])
    got_hits = torch.full_like(want_hits, -7)
    for _ in range(3):                                          # buffer sets (and the mapped peer buffers, the counters) are reused across calls
        sh.lookup9(keys, got_hits, miss_value=0)
    torch.cuda.synchronize()
    assert torch.equal(got_hits, want_hits), "hits differ"
    assert (want_hits != 0).float().mean().item() > 0.5
    for m in modes:
        t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
        s = torch.full_like(t, -1)
        sh.classify9(keys, n_reads, stride, t, s, method=m[0], strategy=m[1], factor=0.25, lower_bound=m[3], fragments_per_read=m[2],
                     min_seed_size=3, max_gap_size=1)
        torch.cuda.synchronize()
        assert torch.equal(s, want[m][1]), (m, "status")
        assert torch.equal(t, want[m][0]), (m, "taxon")
    sh.check()
    sh.close()
    comm.close()
    print("rank %d of %d: %d reads, bit-identical to the whole table" % (rank, world, n_reads), flush=True)


if __name__ == "__main__":
    main()
