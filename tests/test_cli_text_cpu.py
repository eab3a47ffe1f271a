"""The text-only subcommands of the `umgap` command (seedextend incl. --ranked, uniq) against the oracle CLI, byte
for byte.  They never touch the device (in the reference they are plain text filters too:
src/commands/seedextend.rs, src/commands/uniq.rs), so these run on the CPU-only container."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UMGAP = os.path.join(ROOT, "umgap_amd", "bin", "umgap")


def run(args, stdin=b"", **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    p = subprocess.run([UMGAP] + list(args), input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    return p.returncode, p.stdout, p.stderr


@pytest.fixture(scope="module")
def taxonomy(tmp_path_factory):
    if not os.path.exists(UMGAP):
        pytest.skip("umgap_amd/bin/umgap is not built (python -c 'import __graft_entry__ as g; g.build()')")
    from umgap_amd.synth import synthetic_taxonomy, write_taxonomy_tsv
    d = tmp_path_factory.mktemp("text")
    ids, par, valid, rank = synthetic_taxonomy(3000, 5000, max_depth=15, seed=11)
    tsv = str(d / "taxons.tsv")
    write_taxonomy_tsv(tsv, ids, par, valid, rank)
    return {"tsv": tsv, "ids": ids}


def _records(ids, rng, n, unknown=True):
    pool_all = ids.tolist()
    recs = []
    for r in range(n):
        pool = [int(x) for x in rng.choice(pool_all, int(rng.integers(1, 5)))] + [0, 0]
        if unknown and rng.random() < 0.2:
            pool += [9_999_999, 4999]          # beyond the id range / inside it (possibly absent)
        items, i = [], 0
        while i < int(rng.integers(0, 45)):
            t = int(rng.choice(pool))
            run_len = int(rng.integers(1, 5))
            items += [t] * run_len
            i += run_len
        recs.append(">r%d/%d\n" % (r // 2, r % 2 + 1) + "".join("%d\n" % t for t in items))
    return "".join(recs).encode()


@pytest.mark.parametrize("flags", [[], ["-g", "1", "-s", "3"], ["-s", "4", "-g", "2"], ["-s", "1"], ["-g", "3", "-s", "2"]])
def test_seedextend_matches_oracle(taxonomy, flags):
    rng = np.random.default_rng(len(flags) * 7 + 1)
    data = _records(taxonomy["ids"], rng, 600)
    rc, out, err = run(["seedextend"] + flags, data)
    orc, oout, _ = O.run_cli(["seedextend"] + flags, data)
    assert rc == orc and out == oout, err


@pytest.mark.parametrize("flags", [[], ["-p", "0"], ["-g", "2", "-s", "3", "-p", "20"], ["-g", "1", "-s", "2", "-p", "13"]])
def test_seedextend_ranked_matches_oracle(taxonomy, flags):
    """-r taxons.tsv: only the extended seed with the best rank score survives (src/commands/seedextend.rs:150-163);
    ties keep the last one (Iterator::max_by_key), taxa without a ranked ancestor cost the -p penalty."""
    rng = np.random.default_rng(len(flags) * 13 + 2)
    data = _records(taxonomy["ids"], rng, 600)
    rc, out, err = run(["seedextend", "-r", taxonomy["tsv"]] + flags, data)
    orc, oout, _ = O.run_cli(["seedextend", "-r", taxonomy["tsv"]] + flags, data)
    assert rc == orc == 0 and out == oout, err
    plain_flags = list(flags)
    if "-p" in plain_flags:
        i = plain_flags.index("-p")
        del plain_flags[i:i + 2]
    rc2, plain, _ = run(["seedextend"] + plain_flags, data)
    assert rc2 == 0 and out != plain      # the filter does something on this input


@pytest.mark.parametrize("block,threads", [(64, 3), (3000, 8)])
def test_seedextend_ranked_block_seams(taxonomy, block, threads):
    rng = np.random.default_rng(block)
    data = _records(taxonomy["ids"], rng, 300)
    rc, out, _ = run(["seedextend", "-g", "1", "-s", "2", "-r", taxonomy["tsv"]], data, UMGAP_BLOCK_BYTES=block, UMGAP_THREADS=threads)
    orc, oout, _ = O.run_cli(["seedextend", "-g", "1", "-s", "2", "-r", taxonomy["tsv"]], data)
    assert rc == orc == 0 and out == oout


@pytest.mark.parametrize("flags", [[], ["-d", "/"], ["-d", "/", "-s", "+"], ["-w"], ["-d", "r"]])
def test_uniq_matches_oracle(taxonomy, flags):
    rng = np.random.default_rng(len(flags) + 40)
    data = _records(taxonomy["ids"], rng, 400, unknown=False)
    rc, out, err = run(["uniq"] + flags, data)
    orc, oout, _ = O.run_cli(["uniq"] + flags, data)
    assert rc == orc == 0 and out == oout, err


def test_text_filter_errors(taxonomy):
    rc, out, err = run(["seedextend"], b">a\n5\nabc\n")
    orc, oout, oerr = O.run_cli(["seedextend"], b">a\n5\nabc\n")
    assert rc == orc == 1 and out == oout and b"invalid digit found in string" in err
    rc, out, err = run(["seedextend", "-r", "/nonexistent.tsv"], b"")
    assert rc == 1 and err.startswith(b"Error: ")
    rc, out, err = run(["uniq"], b"no header\n")
    assert rc == 1 and b"Expected > at beginning of fasta header." in err


def test_differential_fuzz_text_filters(taxonomy):
    """tools/fuzz_cli.py on the two text filters (fixed seed): random byte soup, product vs oracle."""
    p = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "fuzz_cli.py"), "150", "seedextend", "uniq"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, FUZZ_SEED="3"))
    assert p.returncode == 0, p.stdout[-3000:]
