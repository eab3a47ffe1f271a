"""The `umgap` command (umgap_amd/bin/umgap, C++ over the C ABI) against the oracle CLI, byte for
byte: same stdin, same flags => same stdout and exit status.  Needs the GPU (no CPU fallback)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UMGAP = os.path.join(ROOT, "umgap_amd", "bin", "umgap")
LETTERS = "ACDEFGHIKLMNPQRSTVWY"


def run(args, stdin=b""):
    p = subprocess.run([UMGAP] + list(args), input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)   # (a hang is a failure, not a stuck suite)
    return p.returncode, p.stdout, p.stderr


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    """A 4 000-taxon taxonomy TSV and a 20 000-key 9-mer index (TSV for the oracle, UMGAPIX1 for the product)."""
    from umgap_amd.synth import synthetic_taxonomy, write_taxonomy_tsv
    d = tmp_path_factory.mktemp("cli")
    ids, par, valid, rank = synthetic_taxonomy(4000, 6000, max_depth=20, seed=3)
    tsv = str(d / "taxons.tsv")
    write_taxonomy_tsv(tsv, ids, par, valid, rank)
    rng = np.random.default_rng(0)
    kmers = sorted({"".join(rng.choice(list(LETTERS), 9)) for _ in range(20000)})
    vals = rng.choice(ids, len(kmers))
    idx_tsv = str(d / "index.tsv")
    with open(idx_tsv, "w") as f:
        for k, v in zip(kmers, vals):
            f.write("%s\t%d\n" % (k, v))
    idx = str(d / "ninemer.umgapix")
    with open(idx_tsv, "rb") as fin, open(idx, "wb") as fout:
        p = subprocess.run([UMGAP, "buildindex"], stdin=fin, stdout=fout, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    return {"tsv": tsv, "idx_tsv": idx_tsv, "idx": idx, "kmers": kmers, "ids": ids, "rng": rng}


def test_version():
    rc, out, _ = run(["-V"])
    assert rc == 0 and out == b"umgap 1.1.1\n"   # scripts/umgap-analyse.sh:107


def _peptide_fasta(w, n_records, rng):
    recs = []
    for r in range(n_records):
        lines = [">read%d some description" % r]
        for _ in range(int(rng.integers(0, 12))):
            c = rng.random()
            if c < 0.6:
                lines.append(w["kmers"][int(rng.integers(0, len(w["kmers"])))])
            elif c < 0.8:
                lines.append("".join(rng.choice(list(LETTERS), 9)))
            elif c < 0.9:
                lines.append("".join(rng.choice(list(LETTERS), int(rng.integers(1, 20)))))  # not a 9-mer
            else:
                lines.append("ACD*FGHIK")
        recs.append("\n".join(lines))
    return ("\n".join(recs) + "\n").encode()


@pytest.mark.parametrize("flags", [[], ["-o"], ["-o", "-m", "-c", "12"], ["--one-on-one", "--chunksize=7"]])
def test_pept2lca_matches_oracle(world, flags):
    rng = np.random.default_rng(len(flags))
    fasta = _peptide_fasta(world, 500, rng)
    rc, out, err = run(["pept2lca"] + flags + [world["idx"]], fasta)
    oflags = [f for f in flags if f in ("-o",)] + (["-o"] if "--one-on-one" in flags else [])
    orc, oout, _ = O.run_cli(["pept2lca"] + oflags + [world["idx_tsv"]], fasta)
    assert rc == 0 == orc and out == oout, err


def test_pept2lca_errors(world):
    rc, out, err = run(["pept2lca", world["idx"]], b">a\nAAAAAAAAA\nnot a header? no: an item\n>b\n")
    assert rc == 0
    rc, out, err = run(["pept2lca", world["idx"]], b"oops\n")
    assert rc == 1 and b"Expected > at beginning of fasta header." in err
    rc, out, err = run(["pept2lca", "/nonexistent"], b"")
    assert rc == 1 and err.startswith(b"Error: ")
    rc, out, err = run(["pept2lca"], b"")
    assert rc == 1


@pytest.mark.parametrize("k_flags", [[], ["-o"], ["-o", "-k", "9"]])
def test_prot2kmer2lca_matches_oracle(world, k_flags):
    rng = np.random.default_rng(5)
    recs = []
    for r in range(300):
        n = int(rng.integers(0, 60))
        prot = "".join(rng.choice(list(LETTERS + "*"), n))
        if rng.random() < 0.5 and n > 20:
            s = int(rng.integers(0, n - 12))
            prot = prot[:s] + world["kmers"][int(rng.integers(0, len(world["kmers"])))] + prot[s + 9:]
        # hard-wrapped input is unwrapped by the reader (src/io/fasta.rs:62-64)
        recs.append(">p%d\n%s\n%s" % (r, prot[: n // 2], prot[n // 2:]))
    fasta = ("\n".join(recs) + "\n").encode()
    rc, out, err = run(["prot2kmer2lca"] + k_flags + [world["idx"]], fasta)
    orc, oout, _ = O.run_cli(["prot2kmer2lca"] + k_flags + [world["idx_tsv"]], fasta)
    assert rc == 0 == orc and out == oout, err


@pytest.mark.parametrize("shards,local,procs", [(1, False, False), (2, True, False), (5, True, False), (2, True, True), (3, True, True)])
def test_prot2kmer2lca_shards(world, shards, local, procs):
    """`umgap prot2kmer2lca --shards N`: the index sharded over N GPUs, what the resident server of
    scripts/umgap-analyse.sh:257-264 becomes when the index outgrows one GPU.  Same bytes out as the single-table command.
    A one-GPU box runs N = 1 over RCCL and N > 1 with every rank on GPU 0 (UMGAP_SHARDS_LOCAL).  `procs`: the ranks are
    PROCESSES (`--shard-processes`: this command is rank 0 and starts the others, the exchange is the IPC communicator's
    peer-written one, the batches travel through a shared-memory block) -- the form a node of GPUs runs, rehearsed here with
    the processes sharing the one GPU; a small UMGAP_SHARDS_MAX_KEYS makes a batch take several collective rounds."""
    rng = np.random.default_rng(50 + shards)
    recs = []
    for r in range(2000):
        n = int(rng.integers(0, 70))
        prot = "".join(rng.choice(list(LETTERS + "*"), n))
        if rng.random() < 0.6 and n > 20:
            s = int(rng.integers(0, n - 12))
            prot = prot[:s] + world["kmers"][int(rng.integers(0, len(world["kmers"])))] + prot[s + 9:]
        recs.append(">p%d\n%s\n" % (r, prot))
    fasta = "".join(recs).encode()
    rc0, out0, err0 = run(["prot2kmer2lca", "-o", world["idx"]], fasta)
    env = {"UMGAP_SHARDS_LOCAL": "1"} if local else {}
    extra = ["--shard-processes"] if procs else []
    if procs:
        env["UMGAP_SHARDS_MAX_KEYS"] = "4096"
    rc, out, err = _run_env(["prot2kmer2lca", "-o", "--shards", str(shards)] + extra + [world["idx"]], fasta, UMGAP_BLOCK_BYTES=20000, **env)
    assert rc0 == 0 == rc and out == out0, err
    assert sum(1 for l in out.decode().splitlines() if l.isdigit() and l != "0") > 500
    if procs:
        env.pop("UMGAP_SHARDS_MAX_KEYS")
        env["UMGAP_SHARDS_PROCS"] = "1"          # the switch in the environment, default batch size
        extra = []
    rc, out, err = _run_env(["prot2kmer2lca", "--shards", str(shards)] + extra + [world["idx"]], fasta, **env)
    assert rc == 0 and out == run(["prot2kmer2lca", world["idx"]], fasta)[1], err
    if procs:
        # no worker outlives the command, no shared-memory block either
        ps = subprocess.run(["ps", "-eo", "args"], stdout=subprocess.PIPE).stdout.decode()
        assert "__shard-worker" not in ps
        assert not [f for f in os.listdir("/dev/shm") if f.startswith("umgap-gang-") or f.startswith("umgap-hub-")]


def test_prot2kmer2lca_shard_processes_errors(world, tmp_path):
    """A gang of process-ranks that cannot stand ends with the failing rank's message and exit 1; nothing is left behind."""
    rc, out, err = _run_env(["prot2kmer2lca", "--shards", "2", "--shard-processes", str(tmp_path / "missing.umgapix")], b">a\nAAAAAAAAAA\n",
                            UMGAP_SHARDS_LOCAL="1", UMGAP_IPC_TIMEOUT_S="20")
    assert rc == 1 and err.startswith(b"Error: ") and out == b"", err
    rc, out, err = _run_env(["prot2kmer2lca", "--shards", "65", "--shard-processes", world["idx"]], b"", UMGAP_SHARDS_LOCAL="1")
    assert rc == 1 and b"--shards must be in 1..64" in err
    ps = subprocess.run(["ps", "-eo", "args"], stdout=subprocess.PIPE).stdout.decode()
    assert "__shard-worker" not in ps
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("umgap-gang-")]


def _taxa_fasta(w, rng, n_records, unknown=False):
    ids = w["ids"]
    recs = []
    for r in range(n_records):
        pool = rng.choice(ids, int(rng.integers(1, 8)))
        n = int(rng.integers(0, 40))
        items = rng.choice(pool, n).tolist()
        items = [0 if rng.random() < 0.2 else t for t in items]
        recs.append(">r%d\n" % r + "".join("%d\n" % t for t in items))
    if unknown:
        recs.append(">bad\n%d\n%d\n" % (int(ids[0]), 7_000_001))
        recs.append(">after\n%d\n" % int(ids[1]))
    return "".join(recs).encode()


TAXA2AGG_MODES = [["-a", "lca*"], ["-a", "lca*", "-l", "2"], ["-m", "rmq", "-a", "lca*", "-l2"], ["-m", "rmq", "-a", "mrtl", "-l", "1"],
                  [], ["-a", "hybrid", "-f", "0.5", "-r"], ["-m", "rmq", "-a", "hybrid", "--factor=0.75"], ["--lower-bound", "5", "-a", "lca*", "-r"]]


def test_taxa2agg_taxonomy_with_repeated_rows(world, tmp_path):
    """`taxons.tsv` with rows repeated under the same parent (other rank, other validity): same bytes as the oracle command, which keeps
    the reference's last-row / both-links structures literally."""
    rng = np.random.default_rng(8)
    rows = open(world["tsv"], "rb").read().splitlines()
    extra = []
    for i in rng.choice(len(rows), 12, replace=False):
        f = rows[int(i)].split(b"\t")
        f[2] = rng.choice([b"species", b"genus", b"no rank", b"family"])
        f[4] = rng.choice([b"\x01", b"\x00"])
        extra.append((int(rng.integers(int(i) + 1, len(rows) + 1)), b"\t".join(f)))
    for pos, row in sorted(extra, reverse=True):
        rows.insert(pos, row)
    tsv = tmp_path / "taxons_repeated.tsv"
    tsv.write_bytes(b"\n".join(rows) + b"\n")
    fasta = _taxa_fasta(world, rng, 600)
    for mode in (["-a", "lca*", "-l", "2"], ["-m", "rmq", "-a", "mrtl"], [], ["-r", "-m", "rmq", "-a", "hybrid", "-f", "0.6"]):
        rc, out, err = run(["taxa2agg"] + mode + [str(tsv)], fasta)
        orc, oout, _ = O.run_cli(["taxa2agg"] + mode + [str(tsv)], fasta)
        assert rc == orc and out == oout, (mode, err[-300:])


@pytest.mark.parametrize("mode", TAXA2AGG_MODES)
def test_taxa2agg_matches_oracle(world, mode):
    rng = np.random.default_rng(len(mode) + 11)
    fasta = _taxa_fasta(world, rng, 800)
    rc, out, err = run(["taxa2agg"] + mode + [world["tsv"]], fasta)
    # the oracle CLI takes separated flags only
    omode = []
    for a in mode:
        if a == "-l2":
            omode += ["-l", "2"]
        elif a.startswith("--factor="):
            omode += ["-f", a.split("=")[1]]
        elif a == "--lower-bound":
            omode.append("-l")
        else:
            omode.append(a)
    orc, oout, oerr = O.run_cli(["taxa2agg"] + omode + [world["tsv"]], fasta)
    assert rc == 0 == orc, err
    assert out == oout
    if "rmq" in mode and "hybrid" in mode:
        assert b"Warning: this is a hybrid between LCA/MRTL, not LCA*/MRTL" in err


def test_taxa2agg_error_paths(world):
    rng = np.random.default_rng(99)
    fasta = _taxa_fasta(world, rng, 50, unknown=True)
    for mode, want_rc in ((["-m", "rmq", "-a", "lca*"], 1), (["-a", "lca*"], 101), (["-m", "rmq", "-a", "mrtl"], 101)):
        rc, out, err = run(["taxa2agg"] + mode + [world["tsv"]], fasta)
        orc, oout, oerr = O.run_cli(["taxa2agg"] + mode + [world["tsv"]], fasta)
        assert rc == orc == want_rc, (mode, rc, orc, err)
        assert out == oout          # records before the failing one are printed, nothing after
        if want_rc == 1:
            assert b"Error: Unknown Taxon ID: 7000001" in err
    rc, out, err = run(["taxa2agg", "-m", "tree", "-a", "mrtl", world["tsv"]], b"")
    assert rc == 1 and b"cannot be combined" in err
    rc, out, err = run(["taxa2agg", "-a", "nope", world["tsv"]], b"")
    assert rc == 1
    rc, out, err = run(["taxa2agg", world["tsv"]], b">r\nabc\n")
    assert rc == 1 and b"invalid digit found in string" in err
    rc, out, err = run(["taxa2agg", "/nonexistent.tsv"], b"")
    assert rc == 1 and b"Failed opening taxon file." in err


# Display strings of the reference's taxon ErrorKind values (src/taxon.rs:394-406): Msg carries the bail! text of
# src/taxon.rs:93 / :110; InvalidID is a foreign link to std::num::ParseIntError and UnknownRank to strum::ParseError, whose
# Display texts are those of Rust's std and of strum 0.x.  quick_main! (src/main.rs:8) prints "Error: <display>", exit 1.
TAXON_ERROR_TEXT = {
    "hello world": b"Taxon requires five fields",
    "a\troot\tno_rank\t1\t\x00": b"invalid digit found in string",
    "1\troot\tno_rank\t1\t\x00": b"Matching variant not found",
    "1\troot\tno rank\t#\t\x00": b"invalid digit found in string",
    "1\troot\tno rank\t\t\x00": b"cannot parse integer from empty string",
    "1\troot\tno rank\t7\thello": b"Couldn't parse the valid byte",
}
TAXON_ERROR_KIND = {b"Taxon requires five fields": "Msg", b"Couldn't parse the valid byte": "Msg", b"invalid digit found in string": "InvalidID",
                    b"cannot parse integer from empty string": "InvalidID", b"Matching variant not found": "UnknownRank"}


@pytest.fixture(scope="module")
def proteome_small(fixture_tsv, tmp_path_factory):
    """(index with the single 9-mer ACDEFGHIK -> taxon 2, the reference's fixture taxonomy, the 9-mer, a residue that makes misses)."""
    d = tmp_path_factory.mktemp("small")
    idx = str(d / "one.umgapix")
    p = subprocess.run([UMGAP, "buildindex"], input=b"ACDEFGHIK\t2\n", stdout=open(idx, "wb"), stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    return idx, fixture_tsv, "ACDEFGHIK", "W"


def test_taxon_parsing_vectors_through_the_command(golden, tmp_path):
    """The reference's own taxon-parsing vectors (src/taxon.rs:417-429) fed to the PRODUCT's TSV parser: each line is the
    whole taxons.tsv of a `umgap taxa2agg` run (src/commands/taxa2agg.rs:104 reads it with read_taxa_file, src/taxon.rs:118-128)."""
    g = golden["taxon_parsing"]
    assert len(g["err"]) == 6 and len(g["ok"]) == 4
    for i, (line, kind) in enumerate(g["err"]):
        tsv = tmp_path / ("err%d.tsv" % i)
        tsv.write_bytes(line.encode("latin1") + b"\n")
        want = TAXON_ERROR_TEXT[line]
        assert TAXON_ERROR_KIND[want] == kind            # the message belongs to the ErrorKind the reference's test names
        rc, out, err = run(["taxa2agg", str(tsv)], b">r\n1\n")
        assert rc == 1 and out == b"" and err == b"Error: " + want + b"\n", (line, rc, err)
        orc, oout, oerr = O.run_cli(["taxa2agg", str(tsv)], b">r\n1\n")
        assert (orc, oout, oerr) == (rc, out, err)
    for i, (line, (tid, name, rank, parent, valid)) in enumerate(g["ok"] + [[g["doctest"]["line"], g["doctest"]["want"]]]):
        tsv = tmp_path / ("ok%d.tsv" % i)
        tsv.write_bytes(line.encode("latin1") + b"\n")
        for flags in ([], ["-r"]):
            rc, out, err = run(["taxa2agg"] + flags + [str(tsv)], b">r\n%d\n" % tid)
            if parent == tid:
                # a one-taxon tree: the line parsed, it is its own root (src/taxon.rs:224-247) and the root snaps to itself
                # whatever its valid / rank fields say (filter_ancestors is seeded with Some(root), :279)
                assert rc == 0 and out == b">r\n%d\n" % tid, (line, flags, rc, out, err)
            else:
                # parent != id: parsed, but no taxon is its own parent => `expect("There's no root!")` panics (:243), exit 101
                assert rc == 101 and out == b"" and b"There's no root!" in err, (line, flags, rc, err)
            assert O.run_cli(["taxa2agg"] + flags + [str(tsv)], b">r\n%d\n" % tid)[:2] == (rc, out)


def test_seedextend_slice_panic_exit_101(world, proteome_small):
    """src/commands/seedextend.rs:130-134 + :143-148, sliced at :169-173: with -g1 -s1 the record [0, 5] leaves the scanner
    with start = 2 and end = 1 => `taxons[2..1]` panics, exit 101, nothing of the record is printed (derived by hand from the
    reference source; the oracle agrees).  Both the text filter and the fused command reproduce it."""
    rc, out, err = run(["seedextend", "-g1", "-s1"], b">a\n7\n7\n>b\n0\n5\n>c\n7\n")
    assert rc == 101 and out == b">a\n7\n7\n", (rc, out, err)
    assert O.run_cli(["seedextend", "-g", "1", "-s", "1"], b">a\n7\n7\n>b\n0\n5\n>c\n7\n")[:2] == (101, b">a\n7\n7\n")
    # without the leading miss nothing panics
    rc, out, _ = run(["seedextend", "-g1", "-s1"], b">b\n5\n0\n")
    assert rc == 0 and out == b">b\n5\n"
    # fused command: a fragment whose first window misses and whose second window hits gives seedextend the record [0, x]
    idx, tsv, hit, miss = proteome_small
    frag = miss[0] + hit                      # 10 residues: window 0 = miss[0] + hit[:8] (absent), window 1 = hit
    fasta = (">ok/1\n%s\n>bad/1\n%s\n>after/1\n%s\n" % (hit, frag, hit)).encode()
    stages = [["prot2kmer2lca", "-o", idx], ["seedextend", "-g1", "-s1"], ["uniq", "-d", "/"], ["taxa2agg", "-l1", "-a", "lca*", tsv]]
    data, rcs = fasta, []
    for st in stages:
        rc, data, _ = run(st, data)
        rcs.append(rc)
    assert rcs == [0, 101, 0, 0]              # the reference's pipe: seedextend dies, its readers see EOF after record `ok`
    rc, out, err = run(["prot2kmer2agg", "-g1", "-s1", "-d", "/", "-l1", "-a", "lca*", idx, tsv], fasta)
    assert rc == 101 and out == data, (rc, out, data, err)


def test_text_filters_match_oracle(world, golden):
    g = golden["text_examples"]
    rc, out, _ = run(["uniq"] + g["uniq"]["args"], g["uniq"]["stdin"].encode())
    assert rc == 0 and out == g["uniq"]["stdout"].encode()
    rc, out, _ = run(["seedextend"], g["seedextend"]["stdin"].encode())
    assert rc == 0 and out == g["seedextend"]["stdout"].encode()
    rng = np.random.default_rng(4)
    recs = "".join(">h%d/%d\n%s" % (r // 2, r % 2 + 1, "".join("%d\n" % t for t in rng.choice([0, 0, 5, 5, 7, 9], int(rng.integers(0, 40)))))
                   for r in range(400)).encode()
    for args in (["-g", "1", "-s", "3"], ["-g1", "-s4"], ["--max-gap-size", "2"]):
        oargs = {"-g1": ["-g", "1"], "-s4": ["-s", "4"], "--max-gap-size": ["-g"]}
        flat = []
        for a in args:
            flat += oargs.get(a, [a])
        rc, out, _ = run(["seedextend"] + args, recs)
        orc, oout, _ = O.run_cli(["seedextend"] + flat, recs)
        assert rc == orc == 0 and out == oout
    for args in (["-d", "/"], [], ["-s", ",", "-w"], ["-d", "/", "-s", " "]):
        rc, out, _ = run(["uniq"] + args, recs)
        orc, oout, _ = O.run_cli(["uniq"] + args, recs)
        assert rc == orc == 0 and out == oout


def test_high_precision_pipeline_matches_oracle(world):
    """scripts/umgap-analyse.sh:299-304 minus FGSpp: prot2kmer2lca -o | seedextend -g1 -s3 | uniq -d / | taxa2agg -l2 -a lca*"""
    rng = np.random.default_rng(17)
    recs = []
    for r in range(200):
        for end in (1, 2):
            pep = "".join(rng.choice(list(LETTERS), 50))
            for _ in range(3):  # plant index members so that seeds exist
                s = int(rng.integers(0, 41))
                pep = pep[:s] + world["kmers"][int(rng.integers(0, len(world["kmers"])))] + pep[s + 9:]
            recs.append(">read%d/%d\n%s\n" % (r, end, pep))
    fasta = "".join(recs).encode()

    def pipeline(runner, idx):
        rc, a, _ = runner(["prot2kmer2lca", "-o", idx], fasta)
        assert rc == 0
        rc, b, _ = runner(["seedextend", "-g", "1", "-s", "3"], a)
        assert rc == 0
        rc, c, _ = runner(["uniq", "-d", "/"], b)
        assert rc == 0
        rc, d, _ = runner(["taxa2agg", "-l", "2", "-a", "lca*", world["tsv"]], c)
        assert rc == 0
        return d

    mine = pipeline(run, world["idx"])
    ref = pipeline(O.run_cli, world["idx_tsv"])
    assert mine == ref and mine.count(b">") == 200


@pytest.fixture(scope="module")
def tryptic(tmp_path_factory, world):
    """A tryptic-peptide index: TSV for the oracle, UMGAPIX1 (variable-length table) for the product."""
    d = tmp_path_factory.mktemp("tryp")
    rng = np.random.default_rng(7)
    proteins = []
    for _ in range(400):
        proteins.append("".join(rng.choice(list(LETTERS + "KR"), int(rng.integers(30, 200)))))
    peps = set()
    for prot in proteins:
        cur = ""
        for i, c in enumerate(prot):
            cur += c
            if c in "KR" and i + 1 < len(prot) and prot[i + 1] != "P":
                peps.add(cur)
                cur = ""
        peps.add(cur)
    peps = sorted(p for p in peps if 5 <= len(p) <= 50)
    vals = rng.choice(world["ids"], len(peps))
    tsv = str(d / "tryptic.tsv")
    with open(tsv, "w") as f:
        for k, v in zip(peps, vals):
            f.write("%s\t%d\n" % (k, v))
    idx = str(d / "tryptic.umgapix")
    with open(tsv, "rb") as fin, open(idx, "wb") as fout:
        p = subprocess.run([UMGAP, "buildindex"], stdin=fin, stdout=fout, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    return {"tsv": tsv, "idx": idx, "proteins": proteins, "peps": peps}


def test_pept2lca_tryptic_doc_example(tmp_path):
    """src/commands/pept2lca.rs:24-46 end to end through buildindex."""
    tsv = tmp_path / "t.tsv"
    tsv.write_text("AAALTER\t2\nENFVYLAK\t3398\n")
    idx = tmp_path / "t.umgapix"
    with open(tsv, "rb") as fin, open(idx, "wb") as fout:
        assert subprocess.run([UMGAP, "buildindex"], stdin=fin, stdout=fout).returncode == 0
    rc, out, _ = run(["pept2lca", str(idx)], b">header1\nAAALTER\nENFVYLAK\n")
    assert rc == 0 and out == b">header1\n2\n3398\n"
    rc, out, _ = run(["pept2lca", "-o", str(idx)], b">header1\nNOTATRYPTICPEPTIDE\nENFVYLAK\n")
    assert rc == 0 and out == b">header1\n0\n3398\n"


@pytest.mark.parametrize("flags", [[], ["-o"]])
def test_pept2lca_tryptic_matches_oracle(tryptic, flags):
    rng = np.random.default_rng(3)
    recs = []
    for r in range(400):
        lines = [">r%d" % r]
        for _ in range(int(rng.integers(0, 10))):
            c = rng.random()
            if c < 0.6:
                lines.append(tryptic["peps"][int(rng.integers(0, len(tryptic["peps"])))])
            elif c < 0.9:
                lines.append("".join(rng.choice(list(LETTERS), int(rng.integers(1, 70)))))
            else:
                lines.append("")
        recs.append("\n".join(lines))
    fasta = ("\n".join(recs) + "\n").encode()
    rc, out, err = run(["pept2lca"] + flags + [tryptic["idx"]], fasta)
    orc, oout, _ = O.run_cli(["pept2lca"] + flags + [tryptic["tsv"]], fasta)
    assert rc == 0 == orc and out == oout, err


@pytest.mark.parametrize("flags", [[], ["-o"], ["-l", "9", "-L", "45"], ["-o", "-l9", "-L45"], ["-k", "AC"], ["-d", "W", "-o"]])
def test_prot2tryp2lca_matches_oracle(tryptic, flags):
    """src/commands/prot2tryp2lca.rs:88-140 incl. the settings of scripts/umgap-analyse.sh:291 (-l9 -L45)."""
    rng = np.random.default_rng(9)
    recs = []
    for r in range(300):
        prot = tryptic["proteins"][int(rng.integers(0, len(tryptic["proteins"])))]
        if rng.random() < 0.3:
            s = int(rng.integers(0, len(prot)))
            prot = prot[:s] + "*" + prot[s:]
        if rng.random() < 0.2:
            prot = "KKRPKRRKP" + prot + "K"
        # sequence lines are NOT unwrapped by prot2tryp2lca (Reader::new(stdin, false)): two lines = two digests
        recs.append(">p%d\n%s\n%s\n" % (r, prot[: len(prot) // 2], prot[len(prot) // 2:]))
    fasta = "".join(recs).encode()
    rc, out, err = run(["prot2tryp2lca"] + flags + [tryptic["idx"]], fasta)
    oflags = []
    for a in flags:
        oflags += {"-l9": ["-l", "9"], "-L45": ["-L", "45"]}.get(a, [a])
    orc, oout, _ = O.run_cli(["prot2tryp2lca"] + oflags + [tryptic["tsv"]], fasta)
    assert rc == 0 == orc and out == oout, err
    assert out.count(b"\n") > 600


def test_prot2tryp2lca_custom_pattern(tryptic):
    """-p <regex> (src/commands/prot2tryp2lca.rs:65-68,112-115): the pattern is applied as replace_all(.., "$1\\n$2") twice.  A
    pattern that spells the default differently must reproduce the default's output; a different enzyme (cut after K only) is
    checked against the same two substitutions done in Python over a dict of the index."""
    import re
    rng = np.random.default_rng(19)
    recs, prots = [], []
    for r in range(120):
        prot = tryptic["proteins"][int(rng.integers(0, len(tryptic["proteins"])))]
        if rng.random() < 0.3:
            prot = prot[: len(prot) // 3] + "*" + prot[len(prot) // 3:]
        prots.append(prot)
        recs.append(">p%d\n%s\n" % (r, prot))
    fasta = "".join(recs).encode()
    rc0, out0, _ = run(["prot2tryp2lca", "-o", "-l", "5", "-L", "50", tryptic["idx"]], fasta)
    rc1, out1, err1 = run(["prot2tryp2lca", "-o", "-l", "5", "-L", "50", "-p", "([RK])([^P])", tryptic["idx"]], fasta)
    assert rc0 == 0 == rc1 and out1 == out0, err1
    table = dict(l.split("\t") for l in open(tryptic["tsv"]).read().splitlines())
    # (the pattern in the `regex` crate's spelling, the same in Python's): another enzyme, a class, case-insensitive, named groups
    for rust, py in [("(K)([^P])",) * 2, ("([KRH])([^PD])",) * 2, ("(?i)([kr])([^p])",) * 2, ("(?P<cut>[KR])(?P<next>[^P])",) * 2, (r"([KR])([^P])\z", r"([KR])([^P])\Z")]:
        pat = re.compile(py)
        want = []
        for r, prot in enumerate(prots):
            want.append(">p%d" % r)
            cut = pat.sub(r"\1\n\2", pat.sub(r"\1\n\2", prot)).replace("*", "\n")
            want += [table.get(pep, "0") for pep in cut.split("\n") if 5 <= len(pep) <= 50]
        rc2, out2, err2 = run(["prot2tryp2lca", "-o", "-l", "5", "-L", "50", "--pattern", rust, tryptic["idx"]], fasta)
        assert rc2 == 0 and out2.decode().splitlines() == want, (rust, err2)
        assert (out2 == out0) == (rust.startswith("(?") ), rust      # the two spellings of the default cut like the default
    for refused in ("(K)(?=A)", r"(K)\1"):       # the crate has no look-around and no backreferences
        rc, _, err = run(["prot2tryp2lca", "-p", refused, tryptic["idx"]], fasta)
        assert rc == 1 and err.startswith(b"Error: regex parse error"), err
    rc, _, err = run(["prot2tryp2lca", "-p", "([KR", tryptic["idx"]], fasta)
    assert rc == 1 and err.startswith(b"Error: ")


@pytest.mark.parametrize("k,flags", [(7, ["-o"]), (12, []), (12, ["-o", "-c", "7"]), (64, ["-o"])])
def test_prot2kmer2lca_any_k_on_a_byte_string_index(tryptic, k, flags):
    """src/commands/prot2kmer2lca.rs:71-73,172-179: -k is free; an index whose keys are byte strings serves every window length
    (records shorter than k vanish with their header), byte for byte like the oracle."""
    rng = np.random.default_rng(k)
    peps = [p for p in tryptic["peps"] if len(p) == k]
    recs = []
    for r in range(200):
        prot = tryptic["proteins"][int(rng.integers(0, len(tryptic["proteins"])))][: int(rng.integers(0, 90))]
        if peps and rng.random() < 0.5 and len(prot) > k:
            s = int(rng.integers(0, len(prot) - k))
            prot = prot[:s] + peps[int(rng.integers(0, len(peps)))] + prot[s + k:]
        recs.append(">q%d\n%s\n%s\n" % (r, prot[: len(prot) // 2], prot[len(prot) // 2:]))     # hard-wrapped: unwrapped by the reader
    fasta = "".join(recs).encode()
    rc, out, err = run(["prot2kmer2lca", "-k", str(k)] + flags + [tryptic["idx"]], fasta)
    orc, oout, _ = O.run_cli(["prot2kmer2lca", "-k", str(k)] + flags + [tryptic["tsv"]], fasta)
    assert rc == 0 == orc and out == oout, err
    if k == 12 and "-o" not in flags:
        assert any(l.isdigit() and l != "0" for l in out.decode().splitlines())
    # a multi-byte character inside a window panics like the reference's slice (exit 101), after the completed chunks
    bad = fasta + ">x\nACDEFGHIKLM\xc3\xa9NPQRSTVWYACDEFGHIKLMNPQRSTVWYACDEFGHIKLMNPQRSTVWYACDEFGHIKLMNPQRSTVWY\n>y\nAAAA\n".encode("latin1")
    rc, out, err = run(["prot2kmer2lca", "-k", str(k), "-o", "-c", "50", tryptic["idx"]], bad)
    orc, oout, _ = O.run_cli(["prot2kmer2lca", "-k", str(k), "-o", "-c", "50", tryptic["tsv"]], bad)
    assert rc == orc == 101 and out == oout, (rc, orc, len(out), len(oout))


def test_tryptic_precision_pipeline_matches_oracle(tryptic, world):
    """scripts/umgap-analyse.sh:295-298 minus FGSpp: prot2tryp2lca -l9 -L45 | uniq -d / | taxa2agg -l5 -m rmq -a mrtl
    (and the tryptic-sensitivity variant -l1)."""
    rng = np.random.default_rng(21)
    recs = []
    for r in range(150):
        for end in (1, 2):
            prot = tryptic["proteins"][int(rng.integers(0, len(tryptic["proteins"])))]
            recs.append(">read%d/%d\n%s\n" % (r, end, prot))
    fasta = "".join(recs).encode()
    for lb in ("1", "5"):
        def pipeline(runner, idx):
            rc, a, _ = runner(["prot2tryp2lca", "-l", "9", "-L", "45", idx], fasta)
            assert rc == 0
            rc, b, _ = runner(["uniq", "-d", "/"], a)
            assert rc == 0
            rc, c, _ = runner(["taxa2agg", "-l", lb, "-m", "rmq", "-a", "mrtl", world["tsv"]], b)
            assert rc == 0
            return c
        mine, ref = pipeline(run, tryptic["idx"]), pipeline(O.run_cli, tryptic["tsv"])
        assert mine == ref and mine.count(b">") == 150


@pytest.mark.parametrize("gang", [None, "processes"])
def test_prot2kmer2lca_socket_server(world, tmp_path, gang):
    """`umgap prot2kmer2lca -m -o -s <socket>` as scripts/umgap-analyse.sh:260 starts it; clients talk to it like
    `nc -NU` (send, half-close, read to EOF).  Connections are served sequentially (src/commands/prot2kmer2lca.rs:116-137).
    gang = "processes": the resident server over an index sharded between two process-ranks (`--shards 2 --shard-processes`, both on
    the one GPU): the gang outlives the connections and ends with the server, however that ends."""
    import socket
    import time
    sock = str(tmp_path / "umgap.sock")
    rng = np.random.default_rng(2)
    fastas = []
    for c in range(3):
        recs = []
        for r in range(200):
            pep = "".join(rng.choice(list(LETTERS), int(rng.integers(0, 60))))
            if len(pep) > 20:
                s = int(rng.integers(0, len(pep) - 10))
                pep = pep[:s] + world["kmers"][int(rng.integers(0, len(world["kmers"])))] + pep[s + 9:]
            recs.append(">c%dr%d\n%s\n" % (c, r, pep))
        fastas.append("".join(recs).encode())
    extra = ["--shards", "2", "--shard-processes"] if gang else []
    srv = subprocess.Popen([UMGAP, "prot2kmer2lca", "-m", "-o", "-s", sock] + extra + [world["idx"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, UMGAP_SHARDS_LOCAL="1"))
    try:
        for _ in range(600):
            if os.path.exists(sock):
                break
            time.sleep(0.1)
        assert os.path.exists(sock), srv.stderr.read() if srv.poll() is not None else "socket never appeared"
        for fasta in fastas:
            c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            c.connect(sock)
            c.sendall(fasta)
            c.shutdown(socket.SHUT_WR)
            got = b""
            while True:
                chunk = c.recv(1 << 16)
                if not chunk:
                    break
                got += chunk
            c.close()
            rc, want, _ = run(["prot2kmer2lca", "-o", world["idx"]], fasta)
            assert rc == 0 and got == want and got.count(b">") > 100
    finally:
        srv.terminate()
        out, _ = srv.communicate(timeout=10)
    assert out.startswith(b"Socket created, listening for connections.\n")
    assert out.count(b"Connection accepted. Processing...\n") == 3 and out.count(b"Connection finished succesfully.\n") == 3
    if gang:
        for _ in range(100):                             # (the workers are told of their parent's death by the kernel)
            ps = subprocess.run(["ps", "-eo", "args"], stdout=subprocess.PIPE).stdout.decode()
            if "__shard-worker" not in ps:
                break
            time.sleep(0.1)
        assert "__shard-worker" not in ps
        assert not [f for f in os.listdir("/dev/shm") if f.startswith("umgap-gang-") or f.startswith("umgap-hub-")]


CODON = {}
for i, a in enumerate("TCAG"):
    for j, b in enumerate("TCAG"):
        for k, c in enumerate("TCAG"):
            CODON[a + b + c] = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"[16 * i + 4 * j + k]


def _six_frames(dna):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = "".join(comp.get(c, "N") for c in reversed(dna))
    for strand in (dna, rc):
        for f in range(3):
            yield "".join(CODON.get(strand[i:i + 3], "X") for i in range(f, len(strand) - 2, 3))


def test_config1_testdata_plumbing(tryptic, world, golden):
    """BASELINE.json configs[0]: the reference's own testdata (A1.fq + A2.fq) through the tryptic-precision stages
    (scripts/umgap-analyse.sh:295-298; 6-frame translation stands in for FGSpp, which is not available): plumbing only --
    headers survive, the two ends are joined by `uniq -d /`, one taxon per read pair, identical to the oracle."""
    import json as _json
    with open(os.path.join(ROOT, "tests", "golden", "testdata_A.json")) as f:
        td = _json.load(f)
    recs = []
    for a, b in zip(td["A1"], td["A2"]):
        assert a["header"].endswith("/1") and b["header"].endswith("/2") and a["header"][:-2] == b["header"][:-2]
        for end in (a, b):       # fastq2fasta interleaves the two files; translate -a emits 6 frames per read
            for frame in _six_frames(end["seq"]):
                recs.append(">%s\n%s\n" % (end["header"], frame))
    fasta = "".join(recs).encode()

    def pipeline(runner, idx):
        rc, a, _ = runner(["prot2tryp2lca", "-o", "-l", "9", "-L", "45", idx], fasta)
        assert rc == 0
        rc, b, _ = runner(["uniq", "-d", "/"], a)
        assert rc == 0
        rc, c, _ = runner(["taxa2agg", "-l", "5", "-m", "rmq", "-a", "mrtl", world["tsv"]], b)
        assert rc == 0
        return c
    mine, ref = pipeline(run, tryptic["idx"]), pipeline(O.run_cli, tryptic["tsv"])
    assert mine == ref
    lines = mine.decode().split("\n")
    headers = [l[1:] for l in lines if l.startswith(">")]
    assert len(td["A1"]) == 100
    assert headers == [a["header"][:-2] for a in td["A1"]]        # 100 pairs -> 100 records, header cut at '/'
    assert all(l.isdigit() for l in lines if l and not l.startswith(">"))


def _run_env(args, stdin, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    p = subprocess.run([UMGAP] + list(args), input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)
    return p.returncode, p.stdout, p.stderr


@pytest.mark.parametrize("block,threads", [(64, 3), (1000, 7), (4096, 16), (100000, 1)])
def test_block_seams_and_threads(world, block, threads):
    """The block-parallel tokenizer must not depend on where blocks and thread ranges are cut."""
    rng = np.random.default_rng(block)
    env = dict(UMGAP_BLOCK_BYTES=block, UMGAP_THREADS=threads)
    fasta = _peptide_fasta(world, 300, rng)
    for flags in ([], ["-o"]):
        rc, out, _ = _run_env(["pept2lca"] + flags + [world["idx"]], fasta, **env)
        orc, oout, _ = O.run_cli(["pept2lca"] + flags + [world["idx_tsv"]], fasta)
        assert rc == orc == 0 and out == oout
    prot = "".join(">p%d\n%s\n" % (r, "".join(rng.choice(list(LETTERS), int(rng.integers(0, 80))))) for r in range(300)).encode()
    rc, out, _ = _run_env(["prot2kmer2lca", "-o", world["idx"]], prot, **env)
    orc, oout, _ = O.run_cli(["prot2kmer2lca", "-o", world["idx_tsv"]], prot)
    assert rc == orc == 0 and out == oout
    taxa = _taxa_fasta(world, rng, 300)
    for mode in (["-a", "lca*", "-l", "2"], ["-m", "rmq", "-a", "mrtl"], []):
        rc, out, _ = _run_env(["taxa2agg"] + mode + [world["tsv"]], taxa, **env)
        orc, oout, _ = O.run_cli(["taxa2agg"] + mode + [world["tsv"]], taxa)
        assert rc == orc == 0 and out == oout
    recs = "".join(">h%d/%d\n%s" % (r // 2, r % 2 + 1, "".join("%d\n" % t for t in rng.choice([0, 0, 5, 5, 7, 9], int(rng.integers(0, 40)))))
                   for r in range(300)).encode()
    for cmd in (["seedextend", "-g", "1", "-s", "3"], ["uniq", "-d", "/"], ["uniq"]):
        rc, out, _ = _run_env(cmd, recs, **env)
        orc, oout, _ = O.run_cli(cmd, recs)
        assert rc == orc == 0 and out == oout, cmd


@pytest.mark.parametrize("block", [48, 1 << 20])
def test_line_ending_and_error_quirks_match_oracle(world, block):
    env = dict(UMGAP_BLOCK_BYTES=block, UMGAP_THREADS=4)
    k = world["kmers"]
    ids = [int(x) for x in world["ids"][:6]]
    cases = [
        (["pept2lca", "-o", world["idx"]], ["pept2lca", "-o", world["idx_tsv"]],
         [(">a\r\n%s\r\n%s\r\n>b\r\n%s" % (k[0], k[1], k[2])).encode(),            # CRLF + no final newline
          (">a\n\n\n%s\n>b\n>c\n" % k[3]).encode(),                                 # empty items, empty records
          b">a\n" + k[0].encode() + b"\n>b\n\xff\xfe\n" + k[1].encode() + b"\n>c\n" + k[2].encode() + b"\n",   # bad UTF-8 in an item
          b">a\n" + k[0].encode() + b"\n>b\xff\n" + k[1].encode() + b"\n",           # bad UTF-8 in a header
          b"nope\n>a\n", b"", b">only header", ">héader ü\n%s\n".encode() % k[0].encode()]),
        (["prot2kmer2lca", "-o", world["idx"]], ["prot2kmer2lca", "-o", world["idx_tsv"]],
         [(">a\r\n%sAC\r\nDE%s\r\n>b\nAAAA\n>c\n%s" % (k[0], k[1], k[2] + k[3])).encode(),
          b">a\n" + (k[0] + k[1]).encode() + b"\n>b\nAC\xc3\xa9DEFGHIKLMN\n>c\n" + k[2].encode() + b"\n",   # window cuts a UTF-8 char: panic
          b">a\n" + k[0].encode() + b"\n>b\n\xff\n"]),
        (["taxa2agg", "-a", "lca*", world["tsv"]], ["taxa2agg", "-a", "lca*", world["tsv"]],
         [(">a\r\n%d\r\n%d\r\n>b\r\n%d" % (ids[0], ids[1], ids[2])).encode(),
          (">a\n%d\n>b\n%d\nxyz\n>c\n%d\n" % (ids[0], ids[1], ids[2])).encode(),     # parse error: b is dropped, a printed
          (">a\n%d\n>b\n%d\n\xff\n>c\n%d\n" % (ids[0], ids[1], ids[2])).encode("latin1"),  # bad UTF-8: b printed, then error
          (">a\n%d\n>b\n\n" % ids[0]).encode(), b">a\n+%d\n>b\n-1\n" % ids[3]]),
        (["seedextend", "-s", "2"], ["seedextend", "-s", "2"],
         [b">a\r\n5\r\n5\r\n0\r\n>b\n7\n7", b">a\n5\n5\n>b\n7\nq\n>c\n1\n1\n", b">a\n5\n5\n>b\n7\n\xff\n>c\n1\n1\n"]),
        (["uniq", "-d", "/"], ["uniq", "-d", "/"],
         [b">a/1\r\n5\r\n>a/2\r\n6\r\n>b/1\n7", b">a/1\n\n>a/2\n>b/1\n\n\n>c\n", b">a/1\n5\n>a/2\n\xff\n>b/1\n6\n", b">a\n>a\n>a\n"]),
    ]
    for mine, theirs, inputs in cases:
        chunked = mine[0] in ("pept2lca", "prot2kmer2lca")
        for cflag in ([[], ["-c", "1"], ["-c", "2"], ["-c", "3"]] if chunked else [[]]):
            for data in inputs:
                rc, out, err = _run_env(mine[:1] + cflag + mine[1:], data, **env)
                orc, oout, oerr = O.run_cli(theirs[:1] + cflag + theirs[1:], data)
                assert rc == orc, (mine, cflag, data, rc, orc, err, oerr)
                assert out == oout, (mine, cflag, data, out, oout)


def test_chunk_semantics_on_errors_long_stream(world):
    """An error deep into a stream: everything of completed chunks is printed, nothing of the open one,
    whatever the block size (src/io/fasta.rs:118-137)."""
    k = world["kmers"]
    good = "".join(">r%d\n%s\n%s\n" % (r, k[r % len(k)], k[(7 * r) % len(k)]) for r in range(1000)).encode()
    bad = good + b">bad\n\xff\n>after\n" + k[0].encode() + b"\n"
    for block in (256, 5000, 1 << 20):
        for c in ("240", "7", "1000", "1001", "5000"):
            rc, out, err = _run_env(["pept2lca", "-o", "-c", c, world["idx"]], bad, UMGAP_BLOCK_BYTES=block, UMGAP_THREADS=5)
            orc, oout, _ = O.run_cli(["pept2lca", "-o", "-c", c, world["idx_tsv"]], bad)
            assert rc == orc == 1 and out == oout, (block, c, len(out), len(oout))
    prot = "".join(">p%d\n%s\n" % (r, k[r % len(k)] + k[(3 * r) % len(k)]) for r in range(700)).encode() + b">x\nAC\xc3\xa9DEFGHIKLMN\n>y\nAAAAAAAAAA\n"
    for block in (300, 1 << 20):
        for c in ("240", "1", "701"):
            rc, out, err = _run_env(["prot2kmer2lca", "-o", "-c", c, world["idx"]], prot, UMGAP_BLOCK_BYTES=block, UMGAP_THREADS=3)
            orc, oout, _ = O.run_cli(["prot2kmer2lca", "-o", "-c", c, world["idx_tsv"]], prot)
            assert rc == orc == 101 and out == oout, (block, c, len(out), len(oout))


def test_buildindex_printindex_round_trip(world, tryptic, tmp_path):
    """src/commands/buildindex.rs:20-28: `buildindex < tsv > index; printindex index` gives the TSV back."""
    for kind in (world, tryptic):
        tsv, idx = (kind["idx_tsv"], kind["idx"]) if "idx_tsv" in kind else (kind["tsv"], kind["idx"])
        rc, out, err = run(["printindex", idx])
        assert rc == 0, err
        assert out == open(tsv, "rb").read()
    # the documentation example
    t = tmp_path / "tiny.tsv"
    t.write_text("AAAAA\t2759\nBBBBBB\t9153\n")
    i = tmp_path / "tiny.index"
    with open(t, "rb") as fin, open(i, "wb") as fout:
        assert subprocess.run([UMGAP, "buildindex"], stdin=fin, stdout=fout).returncode == 0
    rc, out, _ = run(["printindex", str(i)])
    assert rc == 0 and out == b"AAAAA\t2759\nBBBBBB\t9153\n"
    # keys the table holds inside its slots (<= 17 letters) next to keys in the heap, in one index: the TSV comes back byte for byte
    rng = np.random.default_rng(4)
    ks = sorted({rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8), int(rng.integers(1, 40))).tobytes() for _ in range(5000)} | {b"A", b"AA", b"Z" * 17, b"Z" * 18, b"a1", b"K*"})
    body = b"".join(k + b"\t" + str(i * 7 + 1).encode() + b"\n" for i, k in enumerate(ks))
    t2, i2 = tmp_path / "mixed.tsv", tmp_path / "mixed.index"
    t2.write_bytes(body)
    with open(t2, "rb") as fin, open(i2, "wb") as fout:
        assert subprocess.run([UMGAP, "buildindex"], stdin=fin, stdout=fout).returncode == 0
    rc, out, _ = run(["printindex", str(i2)])
    assert rc == 0 and out == body
    # out-of-order and duplicate keys are refused like fst::MapBuilder::insert does
    for bad in (b"BBBBB\t1\nAAAAA\t2\n", b"AAAAA\t1\nAAAAA\t2\n", b"AAAAA\tx\n", b"AAAAA\n"):
        p = subprocess.run([UMGAP, "buildindex"], input=bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 1 and p.stderr.startswith(b"Error: ")


@pytest.mark.parametrize("mode", [["-s", "-a", "lca*", "-l", "0.8"], ["-s", "-m", "rmq", "-a", "mrtl"], ["-s", "-a", "hybrid", "-f", "0.5"],
                                  ["-s", "-m", "rmq", "-a", "hybrid", "-l", "0.3"]])
def test_taxa2agg_scored_matches_oracle(world, mode):
    """-s: `taxon=score` items (src/commands/taxa2agg.rs:142-148).  Scores are multiples of 1/64 so every f32 sum is exact
    in any order (the reference's own summation order is HashMap-dependent for the hybrids)."""
    rng = np.random.default_rng(len(mode) * 3 + 1)
    ids = world["ids"]
    recs = []
    for r in range(400):
        pool = rng.choice(ids, int(rng.integers(1, 6)))
        items = ["%d=%s" % (t, repr(float(rng.integers(1, 64)) / 64.0)) for t in rng.choice(pool, int(rng.integers(0, 25)))]
        if rng.random() < 0.2:
            items.append("0=1")
        recs.append(">r%d\n%s" % (r, "".join(i + "\n" for i in items)))
    fasta = "".join(recs).encode()
    rc, out, err = run(["taxa2agg"] + mode + [world["tsv"]], fasta)
    orc, oout, _ = O.run_cli(["taxa2agg"] + mode + [world["tsv"]], fasta)
    assert rc == orc == 0 and out == oout, err
    for bad in (b">r\n5\n", b">r\n5=1=2\n", b">r\n5=abc\n", b">r\nx=1\n"):
        rc, _, err = run(["taxa2agg", "-s", world["tsv"]], bad)
        orc, _, oerr = O.run_cli(["taxa2agg", "-s", world["tsv"]], bad)
        assert rc == orc == 1 and err.split(b"\n")[0] == oerr.split(b"\n")[0]


# ---- prot2kmer2agg: the four-stage 9-mer pipeline as one command -------------------
@pytest.fixture(scope="module")
def proteome(tmp_path_factory, world):
    """An index in which consecutive 9-mers share taxa (all windows of a protein map to its taxon or an ancestor
    of it), so that seedextend finds seeds; one key maps to a taxon the taxonomy does not hold."""
    from umgap_amd.synth import synthetic_taxonomy
    d = tmp_path_factory.mktemp("proteome")
    rng = np.random.default_rng(23)
    ids, par, valid, rank = synthetic_taxonomy(4000, 6000, max_depth=20, seed=3)
    parent_of = dict(zip(ids.tolist(), par.tolist()))
    proteins, table = [], {}
    for _ in range(150):
        prot = "".join(rng.choice(list(LETTERS), int(rng.integers(40, 160))))
        t = int(rng.choice(ids))
        proteins.append(prot)
        for i in range(len(prot) - 8):
            tt = t
            if (i // 7) % 3 == 2:           # stretches annotated with the parent: LCA* has work to do
                tt = parent_of[t]
            table.setdefault(prot[i:i + 9], tt)
    table["WWWWWWWWW"] = 7_000_001           # beyond the taxonomy's id range: parents[id] panics (src/tree/mod.rs:37)
    missing = min(set(range(1, 6000)) - set(ids.tolist()))
    table["YYYYYYYYY"] = missing             # inside the range but absent: UnknownTaxon
    tsv = str(d / "index.tsv")
    with open(tsv, "w") as f:
        for k in sorted(table):
            f.write("%s\t%d\n" % (k, table[k]))
    idx = str(d / "ninemer.umgapix")
    with open(tsv, "rb") as fin, open(idx, "wb") as fout:
        p = subprocess.run([UMGAP, "buildindex"], stdin=fin, stdout=fout, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    return {"idx": idx, "idx_tsv": tsv, "proteins": proteins, "missing": missing}


def _read_fasta(pw, rng, n_reads, crlf=False):
    out = []
    for r in range(n_reads):
        ends = int(rng.choice([1, 2, 2, 2, 3]))
        for e in range(1, ends + 1):
            prot = pw["proteins"][int(rng.integers(0, len(pw["proteins"])))]
            c = rng.random()
            if c < 0.08:
                frag = "".join(rng.choice(list(LETTERS), int(rng.integers(0, 9))))          # shorter than k: dropped
            else:
                s = int(rng.integers(0, max(1, len(prot) - 20)))
                frag = list(prot[s:s + int(rng.integers(12, 70))])
                for i in range(len(frag)):
                    m = rng.random()
                    if m < 0.03:
                        frag[i] = str(rng.choice(list(LETTERS)))
                    elif m < 0.04:
                        frag[i] = "*"
                    elif m < 0.045:
                        frag[i] = "x"
                frag = "".join(frag)
            hdr = ">read%d/%d some text" % (r, e) if rng.random() < 0.9 else ">read%d" % r
            if rng.random() < 0.2 and len(frag) > 4:     # hard-wrapped record
                cut = int(rng.integers(1, len(frag)))
                out.append("%s\n%s\n%s\n" % (hdr, frag[:cut], frag[cut:]))
            else:
                out.append("%s\n%s\n" % (hdr, frag))
    text = "".join(out)
    if crlf:
        text = text.replace("\n", "\r\n")
    return text.encode()


def _oracle_pipe(idx_tsv, tsv, fasta, se, uq, agg):
    rc, a, _ = O.run_cli(["prot2kmer2lca", "-o", idx_tsv], fasta)
    assert rc == 0
    if se is not None:
        rc, a, _ = O.run_cli(["seedextend"] + se, a)
        assert rc == 0
    rc, a, _ = O.run_cli(["uniq"] + uq, a)
    assert rc == 0
    return O.run_cli(["taxa2agg"] + agg + [tsv], a)


FUSED_CASES = [
    (["-g", "1", "-s", "3"], ["-d", "/"], ["-l", "2", "-a", "lca*"]),                    # high precision, umgap-analyse.sh:299-304
    (["-g", "1", "-s", "4"], ["-d", "/"], ["-l", "5", "-a", "lca*"]),                    # max precision :305-310
    (["-g", "1", "-s", "3"], ["-d", "/"], ["-l", "1", "-a", "hybrid", "-f", "0.25"]),    # high sensitivity :283-288
    (["-g", "1", "-s", "2"], ["-d", "/"], ["-l", "1", "-m", "rmq", "-a", "mrtl"]),       # max sensitivity :277-282
    (["-s", "2"], [], ["-a", "lca*", "-r"]),                                            # command defaults, no delimiter
    (None, ["-d", " "], ["-m", "rmq", "-a", "lca*"]),                                   # no seedextend stage
]


@pytest.mark.parametrize("case", range(len(FUSED_CASES)))
def test_prot2kmer2agg_matches_four_stage_pipe(world, proteome, case):
    se, uq, agg = FUSED_CASES[case]
    rng = np.random.default_rng(100 + case)
    fasta = _read_fasta(proteome, rng, 400, crlf=(case == 4))
    orc, ref, _ = _oracle_pipe(proteome["idx_tsv"], world["tsv"], fasta, se, uq, agg)
    assert orc == 0
    flags = (se if se is not None else ["--no-seedextend"]) + uq + agg
    rc, out, err = run(["prot2kmer2agg"] + flags + [proteome["idx"], world["tsv"]], fasta)
    assert rc == 0, err
    assert out == ref
    assert ref.count(b"\n1\n") < ref.count(b">")        # not all reads fell back to the root: seeds existed
    # the same through the product's own four commands
    rc, a, _ = run(["prot2kmer2lca", "-o", proteome["idx"]], fasta)
    if se is not None:
        rc, a, _ = run(["seedextend"] + se, a)
    rc, a, _ = run(["uniq"] + uq, a)
    rc, a, _ = run(["taxa2agg"] + agg + [world["tsv"]], a)
    assert rc == 0 and a == ref


@pytest.mark.parametrize("block,threads", [(64, 3), (700, 5), (5000, 16)])
def test_prot2kmer2agg_block_seams(world, proteome, block, threads):
    """Reads whose records fall into different blocks / thread ranges are still merged (uniq state is carried)."""
    rng = np.random.default_rng(block)
    fasta = _read_fasta(proteome, rng, 300)
    se, uq, agg = FUSED_CASES[0]
    orc, ref, _ = _oracle_pipe(proteome["idx_tsv"], world["tsv"], fasta, se, uq, agg)
    rc, out, err = _run_env(["prot2kmer2agg"] + se + uq + agg + [proteome["idx"], world["tsv"]], fasta, UMGAP_BLOCK_BYTES=block, UMGAP_THREADS=threads)
    assert rc == orc == 0 and out == ref, err


def test_prot2kmer2agg_errors(world, proteome):
    se, uq, agg = FUSED_CASES[0]
    base = ["prot2kmer2agg"] + se + uq + agg
    good = _read_fasta(proteome, np.random.default_rng(5), 50)
    for letter, want_rc in (("Y", 1), ("W", 101)):
        boom = (">boom/1\n%s\n>boom/2\n%s\n>after/1\n%s\n" % (letter * 13, letter * 12, proteome["proteins"][0][:40])).encode()
        orc, ref, oerr = _oracle_pipe(proteome["idx_tsv"], world["tsv"], good + boom, se, uq, agg)
        rc, out, err = run(base + [proteome["idx"], world["tsv"]], good + boom)
        assert rc == orc == want_rc and out == ref          # the reads before the failing one, nothing after
        if want_rc == 1:
            msg = b"Unknown Taxon ID: %d" % proteome["missing"]
            assert msg in err and msg in oerr
    rc, out, err = run(base + [proteome["idx"], world["tsv"]], b"no header\n")
    assert rc == 1 and b"Expected > at beginning of fasta header." in err
    rc, out, err = run(base + [proteome["idx"], world["tsv"]], b"")
    assert rc == 0 and out == b""
    rc, out, err = run(base + [proteome["idx"]], b"")
    assert rc == 1
    rc, out, err = run(["prot2kmer2agg", "-s", "1", proteome["idx"], world["tsv"]], b"")
    assert rc == 0 and out == b""       # --min-seed-size < 2 is served since round 2 (test_seedextend_slice_panic_exit_101 has the panic case)
    rc, out, err = run(["prot2kmer2agg", "-m", "tree", "-a", "mrtl", proteome["idx"], world["tsv"]], b"")
    assert rc == 1 and b"cannot be combined" in err
    # a window edge inside a multi-byte character panics like &seq[i..i+k] does; reads before it are printed
    rc, out, err = run(base + [proteome["idx"], world["tsv"]], good + ">p/1\nACDEFGHIKéLMNPQRST\n".encode())
    orc, ref, _ = _oracle_pipe(proteome["idx_tsv"], world["tsv"], good, se, uq, agg)
    assert rc == 101 and out == ref


def test_prot2kmer2agg_socket_server(world, proteome, tmp_path):
    """`--socket`: index and taxonomy stay resident, clients talk like `nc -NU` (scripts/umgap-analyse.sh:260-264);
    a connection that dies with an error does not take the server down."""
    import socket
    import time
    sock = str(tmp_path / "agg.sock")
    se, uq, agg = FUSED_CASES[0]
    flags = se + uq + agg
    rng = np.random.default_rng(8)
    fastas = [_read_fasta(proteome, rng, 150) for _ in range(2)]
    fastas.insert(1, fastas[0][:2000] + b">boom/1\n" + b"Y" * 13 + b"\n")
    srv = subprocess.Popen([UMGAP, "prot2kmer2agg"] + flags + ["--socket", sock, proteome["idx"], world["tsv"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        for _ in range(600):
            if os.path.exists(sock):
                break
            time.sleep(0.1)
        assert os.path.exists(sock), srv.stderr.read() if srv.poll() is not None else "socket never appeared"
        for fasta in fastas:
            c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            c.connect(sock)
            c.sendall(fasta)
            c.shutdown(socket.SHUT_WR)
            got = b""
            while True:
                chunk = c.recv(1 << 16)
                if not chunk:
                    break
                got += chunk
            c.close()
            rc, want, _ = run(["prot2kmer2agg"] + flags + [proteome["idx"], world["tsv"]], fasta)
            assert got == want and rc == (1 if b"boom" in fasta else 0)
    finally:
        srv.terminate()
        out, _ = srv.communicate(timeout=10)
    assert out.startswith(b"Socket created, listening for connections.\n")
    assert out.count(b"Connection accepted. Processing...\n") == 3 and out.count(b"Connection finished succesfully.\n") == 2
    assert b"Connection died with an error: Unknown Taxon ID: %d\n" % proteome["missing"] in out


def test_prot2kmer2lca_header_only_records(world):
    """A record without sequence lines has the empty sequence (found by tools/fuzz_cli.py: it used to inherit the previous
    record's first line)."""
    k = world["kmers"][0]
    fasta = (">a\n%sACD\n>empty\n>b\n%s\n>also empty\n\n>c\nACDEFGHIKLMN\n" % (k, k)).encode()
    for flags in ([], ["-o"]):
        rc, out, _ = run(["prot2kmer2lca"] + flags + [world["idx"]], fasta)
        orc, oout, _ = O.run_cli(["prot2kmer2lca"] + flags + [world["idx_tsv"]], fasta)
        assert rc == orc == 0 and out == oout and b"empty" not in out


def test_differential_fuzz_device_commands():
    """tools/fuzz_cli.py, a short fixed-seed run: random FASTA-like byte soup through the device-backed commands and the
    fused command, product vs oracle (stdout and exit status)."""
    p = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "fuzz_cli.py"), "20", "taxa2agg", "taxa2agg_scored", "pept2lca", "pept2lca_var", "prot2kmer2lca", "prot2tryp2lca",
                        "prot2kmer2agg"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, FUZZ_SEED="7"))
    assert p.returncode == 0, p.stdout[-3000:]


@pytest.mark.parametrize("block,threads,piped", [(200, 3, True), (4096, 8, False), (1 << 20, 2, True)])
def test_buildindex_streaming(tmp_path, block, threads, piped):
    """buildindex is two block-parallel passes (count + check, then pack + insert through umgap_index_builder_*): the
    result must not depend on block seams, thread ranges or whether stdin is a file (re-read) or a pipe (spooled)."""
    rng = np.random.default_rng(block)
    kmers = sorted({"".join(rng.choice(list(LETTERS), 9)) for _ in range(30_000)})
    tsv = tmp_path / "big.tsv"
    tsv.write_text("".join("%s\t%d\n" % (k, int(rng.integers(1, 1 << 31))) for k in kmers))
    idx = tmp_path / "big.umgapix"
    env = dict(os.environ, UMGAP_BLOCK_BYTES=str(block), UMGAP_THREADS=str(threads))
    with open(idx, "wb") as fout:
        if piped:
            cat = subprocess.Popen(["cat", str(tsv)], stdout=subprocess.PIPE)
            p = subprocess.run([UMGAP, "buildindex"], stdin=cat.stdout, stdout=fout, stderr=subprocess.PIPE, env=env)
            cat.wait()
        else:
            with open(tsv, "rb") as fin:
                p = subprocess.run([UMGAP, "buildindex"], stdin=fin, stdout=fout, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr
    rc, out, err = run(["printindex", str(idx)])
    assert rc == 0 and out == tsv.read_bytes(), err
    # order violations are found wherever the seam falls
    lines = tsv.read_text().splitlines(keepends=True)
    for at in (1, len(lines) // 2, len(lines) - 1):
        for bad in (lines[:at] + [lines[at - 1]] + lines[at:], lines[:at - 1] + [lines[at], lines[at - 1]] + lines[at + 1:]):
            p = subprocess.run([UMGAP, "buildindex"], input="".join(bad).encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert p.returncode == 1 and (b"duplicate key" in p.stderr or b"out-of-order key" in p.stderr), p.stderr


def test_buildindex_proteome_gives_a_walk_order_file_with_the_same_answers(world, tmp_path):
    """`umgap buildindex --proteome proteins.fa` (not in the reference: include/umgap_hip.h, umgap_walk_builder_*): keys and values
    from the TSV on stdin as always, the file laid out in walk order along the proteins.  Every command must print the same bytes
    from it as from the plain file, and as the oracle from the TSV: the proteins hold some of the index's 9-mers (overlapping, so
    walks exist), many 9-mers the index does not hold, wrapped sequence lines, and most of the index's keys never occur in them."""
    rng = np.random.default_rng(11)
    kmers = world["kmers"]
    proteins = []
    for i in range(300):
        parts = []
        for _ in range(int(rng.integers(1, 6))):
            c = rng.random()
            if c < 0.5:
                parts.append(kmers[int(rng.integers(0, len(kmers)))])
            else:
                parts.append("".join(rng.choice(list(LETTERS), int(rng.integers(0, 40)))))
        proteins.append("".join(parts))
    proteins += proteins[:40]                                          # repeated proteins: every key still listed once
    # an index whose keys also cover whole stretches of some proteins (consecutive windows: walks of more than one entry)
    extra = {}
    for prot in proteins[:120]:
        for i in range(0, max(0, len(prot) - 8)):
            if all(c in LETTERS for c in prot[i:i + 9]):
                extra[prot[i:i + 9]] = int(rng.choice(world["ids"]))
    tsv_lines = {}
    with open(world["idx_tsv"]) as f:
        for line in f:
            k, v = line.rstrip("\n").split("\t")
            tsv_lines[k] = int(v)
    tsv_lines.update({k: v for k, v in extra.items() if k not in tsv_lines})
    idx_tsv = str(tmp_path / "index2.tsv")
    with open(idx_tsv, "w") as f:
        for k in sorted(tsv_lines):
            f.write("%s\t%d\n" % (k, tsv_lines[k]))
    fa = str(tmp_path / "proteins.fa")
    with open(fa, "w") as f:
        for i, prot in enumerate(proteins):
            f.write(">sp|P%05d|X\n" % i)
            for o in range(0, len(prot), 60):                          # wrapped at 60 columns like UniProt's files
                f.write(prot[o:o + 60] + "\n")
    plain, walk = str(tmp_path / "plain.ix"), str(tmp_path / "walk.ix")
    for out, args in ((plain, ["buildindex"]), (walk, ["buildindex", "--proteome", fa])):
        with open(idx_tsv, "rb") as fin, open(out, "wb") as fout:
            p = subprocess.run([UMGAP] + args, stdin=fin, stdout=fout, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr
    assert os.path.getsize(walk) >= os.path.getsize(plain) + 8 * len(tsv_lines)
    rc, out, _ = run(["printindex", walk])
    assert rc == 0 and sorted(out.decode().splitlines()) == ["%s\t%d" % (k, tsv_lines[k]) for k in sorted(tsv_lines)]
    reads = "".join(">r%d\n%s\n" % (i, proteins[int(rng.integers(0, len(proteins)))][int(rng.integers(0, 5)):]) for i in range(400)).encode()
    for args in (["prot2kmer2lca", "-o"], ["prot2kmer2lca"], ["prot2kmer2lca", "-m", "-o"]):
        rc0, out0, _ = run(args + [plain], reads)
        rc1, out1, err1 = run(args + [walk], reads)
        orc, oout, _ = O.run_cli(args + [idx_tsv], reads)
        assert rc0 == rc1 == orc == 0 and out0 == out1 == oout, (args, err1)
    pep = _peptide_fasta({"kmers": sorted(tsv_lines)}, 200, rng)
    for flags in ([], ["-o"]):
        rc0, out0, _ = run(["pept2lca"] + flags + [plain], pep)
        rc1, out1, _ = run(["pept2lca"] + flags + [walk], pep)
        assert rc0 == rc1 == 0 and out0 == out1
    rc0, out0, _ = run(["prot2kmer2agg", "-l", "1", plain, world["tsv"]], reads)
    rc1, out1, err1 = run(["prot2kmer2agg", "-l", "1", walk, world["tsv"]], reads)
    assert rc0 == rc1 == 0 and out0 == out1 and len(out0) > 0, err1
    rc, _, err = run(["buildindex", "--proteome", "/nonexistent"], b"")
    assert rc == 1 and err.startswith(b"Error: ")

