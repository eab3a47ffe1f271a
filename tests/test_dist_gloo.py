"""The N > 1 plumbing on CPU (world size 2, gloo).  The data path of the sharded mode lives in libumgap_hip.so
(umgap_amd/csrc/shard.hip: RCCL send/recv groups) and is checked on the GPU with all ranks on one card
(tests/test_gpu_parity.py::test_sharded_table_*); what can and must work without a GPU is everything around it:
`python bench.py --gpus 2` starting its own ranks before touching the GPU, the rendezvous on 127.0.0.1, ranks_seen,
the hand-off of rank 0's unique id, barriers, the max over ranks, ONE JSON line from rank 0, and a non-zero exit when a
rank fails or stalls."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launch_world2_gloo():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse-cpu"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["steps"] == 3 and j["value"] is None and "rehearsal" in j
    assert j["ms_per_step"] >= 2.0                     # the max over ranks: rank 1 sleeps 2 ms per step, rank 0 only 1


def test_spawn_ranks_reports_failure_and_stall(tmp_path):
    from umgap_amd.dist import spawn_ranks
    bad = tmp_path / "bad.py"
    bad.write_text(textwrap.dedent("""
        import os, sys
        sys.exit(7 if os.environ["RANK"] == "1" else 0)
    """))
    assert spawn_ranks(str(bad), [], 2, timeout=60) == 7
    hang = tmp_path / "hang.py"
    hang.write_text(textwrap.dedent("""
        import os, time
        assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1"
        time.sleep(600 if os.environ["RANK"] == "2" else 0)
    """))
    assert spawn_ranks(str(hang), [], 3, timeout=3) != 0      # the stalled rank is killed and reported


def test_owner_ranges_cover_the_table_once():
    """Shard r of n owns the home buckets [nb*r/n, nb*(r+1)/n) (alloc_table in index.hip = owner_of_bucket in shard.hip):
    a partition of [0, nb) for every n, so every key has exactly one owner."""
    for nb in (1 << 17, 3 << 17, 12345 << 17):
        for n in (1, 2, 3, 5, 8):
            edges = [nb * r // n for r in range(n + 1)]
            assert edges[0] == 0 and edges[-1] == nb and all(a < b for a, b in zip(edges, edges[1:]))
            for b in (0, 1, nb // 3, nb // 2, nb - 1) + tuple(e for e in edges[1:-1]) + tuple(e - 1 for e in edges[1:]):
                s = b * n // nb
                while nb * s // n > b:
                    s -= 1
                while nb * (s + 1) // n <= b:
                    s += 1
                assert edges[s] <= b < edges[s + 1]
