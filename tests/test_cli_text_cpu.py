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
    p = subprocess.run([UMGAP] + list(args), input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)
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


@pytest.mark.parametrize("flags,panics", [(["-s", "1", "-g", "1"], True), (["-s", "1", "-g", "1", "-p", "0"], False),
                                          (["-s", "1", "-g", "1", "-p", "1"], True), (["-s", "0", "-g", "2", "-p", "0"], False)])
def test_seedextend_ranked_inverted_seed(taxonomy, flags, panics):
    """A seed with start > end (src/commands/seedextend.rs:119: a leading zero-run <= g, one taxon, a zero-run > g, under -s 1) in
    ranked mode: `.take(e - s)` (:158) underflows.  A release build wraps -- the seed then scores every taxon from s on, beats the
    real seed by its gap penalties and `taxons[2..1]` panics (exit 101); with -p 0 the two tie, max_by_key keeps the last one and
    `5 5 5` is printed.  (A build with overflow checks panics at the subtraction either way: the oracle's --overflow-checks.)"""
    five = int(taxonomy["ids"][5])
    rec = (">a\n%d\n%d\n%d\n" % (five, five, five) + ">r\n0\n7\n0\n0\n0\n%d\n%d\n%d\n" % (five, five, five) + ">z\n%d\n%d\n" % (five, five)).encode()
    args = ["seedextend", "-r", taxonomy["tsv"]] + flags
    rc, out, err = run(args, rec)
    orc, oout, _ = O.run_cli(args, rec)
    assert rc == orc and out == oout, (rc, orc, out, oout, err)
    assert rc == (101 if panics else 0)
    if panics:
        assert out.startswith(b">a\n") and b">r" not in out              # the records before the panic are out, nothing after
        assert b"panicked" in err
    else:
        assert out.endswith(b">r\n%d\n%d\n%d\n>z\n%d\n%d\n" % ((five,) * 5))
    drc, _, _ = O.run_cli(args + ["--overflow-checks"], rec)
    assert drc == 101                                                     # a debug build dies at `e - s` whatever the penalty
    # more of them: random records rich in short zero-runs, product and oracle record for record (the command stops at its first panic)
    rng = np.random.default_rng(9)
    pool = [five, int(taxonomy["ids"][9]), 0, 0, 0, 7]
    n_inverted_survivors = 0
    for _ in range(300):
        items = [int(rng.choice(pool)) for _ in range(int(rng.integers(1, 14)))]
        data = (">q\n" + "".join("%d\n" % t for t in items)).encode()
        rc, out, _ = run(args, data)
        orc, oout, _ = O.run_cli(args, data)
        assert rc == orc and out == oout, (items, rc, orc, out, oout)
        drc, dout, _ = O.run_cli(args + ["--overflow-checks"], data)
        n_inverted_survivors += (drc == 101 and orc == 0)
    if not panics:
        assert n_inverted_survivors > 0                                   # records where release and debug builds part ways


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


def _big_hits(n_records, seed=3):
    rng = np.random.default_rng(seed)
    vals = rng.integers(0, 50, (n_records, 30))
    vals[rng.random(vals.shape) < 0.3] = 0
    return b"".join(b">r%d/%d\n" % (i // 2, i % 2 + 1) + b"\n".join(b"%d" % v for v in vals[i]) + b"\n" for i in range(n_records))


def test_async_io_same_bytes_through_a_pipe_chain(tmp_path):
    """stdin / stdout behind reader and writer threads (async_io.hpp) against the plain path (UMGAP_ASYNC_IO=0): a chain of two
    stages through a real pipe, blocks much smaller than the data, must give the same bytes -- and everything written must be out
    when the process ends, also when it ends with an error."""
    if not os.path.exists(UMGAP):
        pytest.skip("umgap_amd/bin/umgap is not built")
    data = _big_hits(60_000)
    src = tmp_path / "hits.fa"
    src.write_bytes(data)
    outs = {}
    for mode in ("0", "1"):
        env = dict(os.environ, UMGAP_ASYNC_IO=mode, UMGAP_BLOCK_BYTES=str(1 << 16))
        with open(src, "rb") as fi:
            a = subprocess.Popen([UMGAP, "seedextend", "-g1", "-s2"], stdin=fi, stdout=subprocess.PIPE, env=env)
            b = subprocess.Popen([UMGAP, "uniq", "-d", "/"], stdin=a.stdout, stdout=subprocess.PIPE, env=env)
            a.stdout.close()
            outs[mode] = b.communicate(timeout=120)[0]
            assert a.wait(timeout=60) == 0 and b.returncode == 0
    assert outs["0"] == outs["1"] and len(outs["1"]) > 100_000
    # an error in the middle of the stream: what was written before it is all there, then exit 1
    bad = data[:len(data) // 2] + b">x\nnot-a-number\n" + data[len(data) // 2:]
    res = {m: run(["seedextend", "-g1", "-s2"], bad, UMGAP_ASYNC_IO=m, UMGAP_BLOCK_BYTES=1 << 16) for m in ("0", "1")}
    assert res["0"][0] == res["1"][0] == 1 and res["0"][1] == res["1"][1] and res["0"][2] == res["1"][2] and len(res["1"][1]) > 10_000


def test_async_io_consumer_that_leaves_early(tmp_path):
    """A reader that closes the pipe after a few bytes (`| head`): the stage must end promptly (SIGPIPE), not hang on its writer."""
    if not os.path.exists(UMGAP):
        pytest.skip("umgap_amd/bin/umgap is not built")
    src = tmp_path / "hits.fa"
    src.write_bytes(_big_hits(200_000))
    with open(src, "rb") as fi:
        a = subprocess.Popen([UMGAP, "uniq", "-d", "/"], stdin=fi, stdout=subprocess.PIPE)
        first = a.stdout.read(100)
        a.stdout.close()
        rc = a.wait(timeout=60)
    assert len(first) == 100 and rc != 0


def test_fused_command_tokenizer_against_line_by_line_restatement(tmp_path):
    """The block tokenizer of `umgap prot2kmer2agg` (umgap_amd/csrc/host/fuse_parse.hpp: 64 bytes at a time off newline and
    high-bit masks) against a plain line-by-line restatement of the rules of src/io/fasta.rs:38-67 + src/commands/
    prot2kmer2lca.rs:172-179 on random FASTA-like byte soup (CRs, empty lines, lines around the 64-byte stretches, valid and
    broken UTF-8, a missing last newline), built with AddressSanitizer + UBSan (tests/native/fuse_parse_check.cpp)."""
    import shutil
    import subprocess
    cxx = shutil.which("g++") or shutil.which("clang++")
    if not cxx:
        pytest.skip("no C++ compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fuse_parse_check")
    subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-o", exe, os.path.join(root, "tests", "native", "fuse_parse_check.cpp")], check=True)
    p = subprocess.run([exe, "6000", "11"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()[-2000:]


def test_oracle_lookup_stage_on_threads_prints_the_same_bytes(tmp_path):
    """The CPU twin of the end-to-end measurement looks its chunks up on several threads (oracle CLI `prot2kmer2lca -j N`, the
    stand-in for the reference's rayon par_bridge, src/commands/prot2kmer2lca.rs:163-192) and keeps input order: same bytes as the
    plain loop, errors and panics included."""
    rng = np.random.default_rng(3)
    letters = b"ACDEFGHIKLMNPQRSTVWY"
    idx = tmp_path / "ix.tsv"
    kmers = sorted({bytes(rng.choice(list(letters), 9).tolist()) for _ in range(300)})
    idx.write_bytes(b"".join(k + b"\t%d\n" % (i + 2) for i, k in enumerate(kmers)))
    recs = []
    for r in range(700):
        n = int(rng.choice([0, 5, 9, 30, 60]))
        seq = bytearray(rng.choice(list(letters), n).tolist())
        if n >= 30:
            seq[3:12] = kmers[int(rng.integers(0, len(kmers)))]
        recs.append(b">r%d\n" % r + bytes(seq) + b"\n")
    good = b"".join(recs)
    for text in (good, good + b">bad\nACDEFGHIK\xffACD\nAAAAAAAAAAAA\n", good[:9000] + ">x\nACéDEFGHIKLMNP\n".encode() + good[9000:]):
        a = O.run_cli(["prot2kmer2lca", "-o", "-c", "7", str(idx)], text)
        b = O.run_cli(["prot2kmer2lca", "-o", "-c", "7", "-j", "4", str(idx)], text)
        assert a[0] == b[0] and a[1] == b[1], (a[0], b[0], len(a[1]), len(b[1]))
    assert len(O.run_cli(["prot2kmer2lca", "-o", "-j", "3", str(idx)], good)[1]) > 5000


def test_threaded_host_code_under_thread_sanitizer(tmp_path):
    """The threaded host code of the command -- the reader / writer threads behind stdin / stdout (async_io.hpp, with their buffer
    pool), the block worker pool (fastpath.hpp) and the two overlapped stages -- built with ThreadSanitizer and driven through
    `seedextend` and `uniq` (the commands that need no GPU) over files and over pipes, small blocks so that every seam and hand-over
    is crossed many times.  Any report fails the test (TSAN_OPTIONS exitcode)."""
    import shutil
    import subprocess
    cxx = shutil.which("g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "umgap_amd", "libumgap_hip.so")
    if not cxx or not os.path.exists(lib):
        pytest.skip("needs g++ and the built library")
    exe = str(tmp_path / "umgap_tsan")
    p = subprocess.run([cxx, "-fsanitize=thread", "-O1", "-g", "-std=c++17", "-o", exe, os.path.join(root, "umgap_amd", "csrc", "host", "main.cpp"),
                        "-L" + os.path.dirname(lib), "-lumgap_hip", "-Wl,-rpath," + os.path.dirname(lib), "-lpthread"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if p.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for g++ here: " + p.stdout.decode()[-300:])
    rng = np.random.default_rng(17)
    recs = []
    for r in range(30_000):
        t = int(rng.integers(1, 3_000_000))
        vals = np.where(rng.random(42) < 0.6, t, np.where(rng.random(42) < 0.5, 0, rng.integers(1, 3_000_000, 42)))
        recs.append(b">read%d/%d\n" % (r // 2, r % 2 + 1) + b"\n".join(b"%d" % v for v in vals) + b"\n")
    text = b"".join(recs)
    env = dict(os.environ, UMGAP_BLOCK_BYTES="200000", UMGAP_THREADS="5", TSAN_OPTIONS="exitcode=66 halt_on_error=0")
    ref_env = dict(os.environ, UMGAP_BLOCK_BYTES="200000")
    prod = os.path.join(root, "umgap_amd", "bin", "umgap")
    want_se = subprocess.run([prod, "seedextend", "-g1", "-s3"], input=text, stdout=subprocess.PIPE, env=ref_env, check=True).stdout
    want_un = subprocess.run([prod, "uniq", "-d", "/"], input=want_se, stdout=subprocess.PIPE, env=ref_env, check=True).stdout
    src = tmp_path / "in.fa"
    src.write_bytes(text)
    with open(src, "rb") as fi:     # a file on stdin (mapped), a pipe between the two stages, a pipe out
        p1 = subprocess.Popen([exe, "seedextend", "-g1", "-s3"], stdin=fi, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        p2 = subprocess.Popen([exe, "uniq", "-d", "/"], stdin=p1.stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        p1.stdout.close()
        out, err2 = p2.communicate()
        err1 = p1.stderr.read()
        p1.wait()
    assert p1.returncode == 0 and p2.returncode == 0, (p1.returncode, p2.returncode, err1[-2000:], err2[-2000:])
    assert out == want_un
    p3 = subprocess.run([exe, "seedextend", "-g1", "-s3"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)   # a pipe in
    assert p3.returncode == 0 and p3.stdout == want_se, p3.stderr[-2000:]


def test_cleavage_pattern_rewriter_against_python_re(tmp_path):
    """`umgap prot2tryp2lca -p <regex>` (src/commands/prot2tryp2lca.rs:65-68,96,112-115): the pattern is written in the syntax of
    Rust's `regex` crate and applied as replace_all(.., "$1\\n$2") twice.  umgap_amd/csrc/host/cleave_pattern.hpp rewrites it for
    std::regex; on the subset both read, Python's `re` reads it the same way and is the checker here (ASan + UBSan build).
    What the crate refuses (look-around, backreferences) and what this build cannot express must be refused, not misread."""
    import re
    import shutil
    import subprocess
    cxx = shutil.which("g++") or shutil.which("clang++")
    if not cxx:
        pytest.skip("no C++ compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cleave_pattern_check")
    subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-o", exe, os.path.join(root, "tests", "native", "cleave_pattern_check.cpp")], check=True)
    # (the crate's spelling, Python's spelling of the same pattern)
    good = [("([KR])([^P])",) * 2, ("([RK])([^P])",) * 2, ("(K)([^P])",) * 2, ("([KR])(.)",) * 2, ("([KRH])([^PD])",) * 2, ("(K|R)([^P])",) * 2,
            ("(?i)([kr])([^p])",) * 2, ("(?P<a>[KR])(?P<b>[^P])",) * 2, ("(?<a>[KR])(?<b>[^P])", "(?P<a>[KR])(?P<b>[^P])"), ("K",) * 2, ("(K)",) * 2,
            (r"\A(M)(.)",) * 2, (r"([KR])([^P])\z", r"([KR])([^P])\Z"), ("(?:K|R)([^P])",) * 2, ("([[:upper:]])(P)", "([A-Z])(P)"), (r"(K)(\x{41})", r"(K)(\x41)"),
            ("(K+?)(A)",) * 2, ("(K{2,3})([^P])",) * 2, ("([FWY])([^P])",) * 2, ("([]K])(A)", r"([\]K])(A)"), (r"(\w)(\*)",) * 2, ("(?m)^(M)(.)",) * 2,
            ("([^A-J])([L-Z])",) * 2, ("(D)()",) * 2, ("()(D)",) * 2,
            # escapes: the crate's other spellings of a code point, the bell, its escapable punctuation; a dot that must not stop at \r
            (r"(K)(\u{41})", r"(K)(\x41)"), (r"(K)(\u0041)", r"(K)(\x41)"), (r"(K)(\U00000041)", r"(K)(\x41)"), (r"(\a)(K)", r"(\x07)(K)"), (r"(\*)(\-|\~|\#|\&|A)", r"(\*)(-|~|#|&|A)"),
            ("(K)(.)\r", "(?s)(K)([^\n])\r")]
    bad = [r"(K)(\0)", r"(K)(\e)", r"(\q)(A)", r"(K)(\Q)", r"(K)(?i:a)", "(K)([a[bc]])", r"(K)(\u{1F600})", r"(K)(\u00e9)", "(K)(?=A)", "(K)(?!P)", "(?<=K)(A)", r"(K)\1", "([KR", "(K", r"(\pL)(A)", "(?s)(K)(.)", "(?x)(K) (.)", "(K)([a-z&&[^p]])", "(K)((?i)a)", "(K)(A)\\", "*K"]
    rng = np.random.default_rng(23)
    letters = "ACDEFGHIKLMNPQRSTVWY"
    seqs = ["", "K", "KR", "KP", "AKRPKAAKKRA", "MMKA", "A]AKA*KKKA", "kraKRPkp", "AK\x07KA*-K~", "AKA\rKK\rA"]
    seqs += ["".join(rng.choice(list(letters + "KRKRP*"), int(rng.integers(1, 80)))) for _ in range(60)]
    feed, want = [], []
    for rust, py in good:
        pat = re.compile(py)
        for s in seqs:
            feed.append("%s\t%s\n" % (rust, s))
            want.append(pat.sub(lambda m: (m.group(1) if pat.groups >= 1 else "") + "|" + (m.group(2) if pat.groups >= 2 else ""),
                                pat.sub(lambda m: (m.group(1) if pat.groups >= 1 else "") + "\n" + (m.group(2) if pat.groups >= 2 else ""), s)).replace("\n", "|"))
    for rust in bad:
        feed.append("%s\tAKA\n" % rust)
        want.append(None)
    p = subprocess.run([exe], input="".join(feed).encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = p.stdout.decode().split("\n")[:-1]
    assert len(got) == len(want)
    for line, g, w in zip(feed, got, want):
        if w is None:
            assert g.startswith("ERROR regex parse error:"), (line, g)
        else:
            assert g == w, (line, g, w)
    # a greedy pattern over a long protein: std::regex recurses once per residue; the cut runs on a stack sized for the sequence
    # (the sequences end in KK so that both passes match at their first attempt: a greedy pattern that fails is quadratic in any engine)
    long_seq = "A" * 150_000 + "KK"
    lines = [("(.*)([KR])", long_seq), ("([KR])([^P])", "AC" * 100_000 + "KA" + "C" * 100_000)]
    p = subprocess.run([exe], input="".join("%s\t%s\n" % ln for ln in lines).encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = p.stdout.decode().split("\n")[:-1]
    for (pat_s, seq), g in zip(lines, got):
        pat = re.compile(pat_s)
        assert g == pat.sub(r"\1|\2", pat.sub(r"\1\n\2", seq)).replace("\n", "|"), pat_s


def test_seedextend_and_uniq_on_a_stream_of_many_blocks(tmp_path):
    """The text filters on a stream big enough for their block machinery to turn over several times -- blocks cut ahead of the
    caller by the filler thread (piped stdin), output buffers handed to the writer thread without a copy and reused two blocks later,
    uniq's output flushed in 4 MB pieces while a group is open across the flush -- byte for byte like the oracle's commands."""
    import subprocess
    rng = np.random.default_rng(77)
    recs = []
    taxa = [0, 0, 7, 7, 7, 12, 12, 9, 1, 433, 433, 433, 433]
    for r in range(220_000):
        hdr = b">r%d/%d" % (r // 2, 1 + r % 2)
        n = int(rng.integers(0, 40))
        items = rng.choice(taxa, n)
        recs.append(hdr + b"\n" + b"".join(b"%d\n" % t for t in items))
    data = b"".join(recs)
    assert len(data) > 12 << 20
    env = dict(os.environ, UMGAP_BLOCK_BYTES=str(1 << 20), UMGAP_THREADS="4")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    umgap = os.path.join(root, "umgap_amd", "bin", "umgap")
    if not os.path.exists(umgap):
        pytest.skip("the umgap command is not built")
    for args in (["seedextend", "-g", "1", "-s", "3"], ["uniq", "-d", "/"], ["uniq"]):
        p = subprocess.run([umgap] + args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        orc, oout, _ = O.run_cli(args, data)
        assert p.returncode == orc == 0 and p.stdout == oout, (args, p.returncode, len(p.stdout), len(oout), p.stderr[-300:])


@pytest.mark.parametrize("cmd", [["uniq"], ["seedextend"]])
def test_failed_write_ends_the_command_with_the_io_error(taxonomy, cmd):
    """A write that fails ends every command of the reference with the io error and exit 1 (`writer.write_record(..)?`,
    src/commands/uniq.rs:71-80, seedextend.rs:166-176): a full device, and a reader that hangs up (Rust ignores SIGPIPE: EPIPE, not a
    signal).  Here the writer thread notes the failed write(), the block loop stops at its next block and main reports it."""
    rng = np.random.default_rng(3)
    data = _records(taxonomy["ids"], rng, 4000, unknown=False)
    with open("/dev/full", "wb") as full:
        p = subprocess.run([UMGAP] + cmd, input=data, stdout=full, stderr=subprocess.PIPE)
    assert p.returncode == 1 and p.stderr.strip() == b"Error: No space left on device (os error 28)", (p.returncode, p.stderr)
    big = data * 40                                   # several blocks at 1 MB: the loop must stop, not format the rest for nobody
    pr = subprocess.Popen([UMGAP] + cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, UMGAP_BLOCK_BYTES=str(1 << 20)))
    import threading
    feeder = threading.Thread(target=lambda: (_feed(pr.stdin, big)))
    feeder.start()
    assert pr.stdout.read(16).startswith(b">")
    pr.stdout.close()                                 # the reader hangs up
    err = pr.stderr.read()
    feeder.join()
    assert pr.wait() == 1 and err.strip() == b"Error: Broken pipe (os error 32)", (pr.returncode, err)


def _feed(pipe, data):
    try:
        pipe.write(data)
        pipe.close()
    except (BrokenPipeError, OSError):
        pass


def test_shard_processes_without_a_gpu_fail_loudly_and_leave_nothing(tmp_path):
    """`umgap prot2kmer2lca --shards N --shard-processes` starts its worker ranks before it touches the device.  Where the gang cannot
    stand (here: no GPU, or no index file) the command ends with `Error: ...` and exit 1, no worker outlives it and no shared-memory
    block stays in /dev/shm; the hidden worker subcommand refuses to be started by hand."""
    import time
    if not os.path.exists(UMGAP):
        pytest.skip("umgap_amd/bin/umgap is not built")
    idx = tmp_path / "none.umgapix"
    t0 = time.time()
    rc, out, err = run(["prot2kmer2lca", "--shards", "3", "--shard-processes", str(idx)], b">a\nACDEFGHIKLM\n", UMGAP_SHARDS_LOCAL="1", UMGAP_IPC_TIMEOUT_S="10")
    assert rc == 1 and out == b"" and err.startswith(b"Error: "), (rc, out, err)
    assert time.time() - t0 < 60
    rc, out, err = run(["prot2kmer2lca", "--shards", "0", "--shard-processes", str(idx)], b"")
    assert rc == 1 and b"--shards must be in 1..64" in err
    rc, out, err = run(["__shard-worker"], b"")
    assert rc == 1 and b"is started by" in err
    rc, out, err = run(["__shard-worker", str(idx), "1", "2", "0", "/umgap-hub-none", "/umgap-gang-none", "1"], b"")
    assert rc == 1 and err.startswith(b"Error: "), err
    for _ in range(50):                                  # (a worker notices its parent's end within a nap)
        ps = subprocess.run(["ps", "-eo", "args"], stdout=subprocess.PIPE).stdout.decode()
        if "__shard-worker" not in ps:
            break
        time.sleep(0.1)
    assert "__shard-worker" not in ps
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("umgap-gang-") or f.startswith("umgap-hub-")]


@pytest.mark.parametrize("ranked", [False, True])
def test_seedextend_panic_prints_nothing_of_the_records_behind_it(taxonomy, ranked):
    """A record whose seed is a slice out of range (`0 213` under -s1 -g1: src/commands/seedextend.rs:154-160 / :125) ends the
    reference with a panic when it is written; the records before it are out, nothing of those behind it -- not even the header of
    the record whose `>` line made the filter notice (found by tools/fuzz_cli.py's ranked -s 1 cases, round 5)."""
    flags = ["-s", "1", "-g", "1"] + (["-r", taxonomy["tsv"]] if ranked else [])
    for before in (b"", b">ok\n7\n7\n"):
        fasta = before + b">r0/1\n0\n213\n>x y/z\n>\n5\n>later\n9\n"
        for env in ({}, {"UMGAP_BLOCK_BYTES": "100", "UMGAP_THREADS": "3"}, {"UMGAP_BLOCK_BYTES": "8", "UMGAP_THREADS": "2"}):
            rc, out, err = run(["seedextend"] + flags, fasta, **env)
            orc, oout, _ = O.run_cli(["seedextend"] + flags, fasta)
            assert (rc, out) == (orc, oout) and rc == 101, (env, rc, out, orc, oout, err)
            assert b"x y/z" not in out and b"later" not in out
