#!/usr/bin/env python3
"""Differential fuzzer: the `umgap` command against the oracle CLI on random FASTA-like byte soup (stdout and exit
status must agree).  Text-only commands run anywhere; the device-backed ones need the GPU box.
    python tools/fuzz_cli.py <cases> [seedextend uniq taxa2agg pept2lca prot2kmer2lca prot2kmer2agg]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O

UMGAP = os.environ.get("UMGAP_BIN") or os.path.join(ROOT, "umgap_amd", "bin", "umgap")   # UMGAP_BIN: e.g. a sanitizer build of the host code
LETTERS = "ACDEFGHIKLMNPQRSTVWY"


def run(args, data, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([UMGAP] + args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    run.last_stderr = p.stderr
    return p.returncode, p.stdout


def soup(rng, items, n_records, nl_choices=(b"\n", b"\n", b"\n", b"\r\n")):
    out = []
    for r in range(n_records):
        c = rng.random()
        hdr = b">" + rng.choice([b"r%d/1" % (r // 2), b"r%d/2" % (r // 2), b"", b"x y/z", b">>", b"r\r", b"h\xc3\xa9"])
        if c < 0.03:
            hdr = rng.choice([b"", b"junk", b" >late"])          # not a header at all
        out.append(hdr + rng.choice(nl_choices))
        for _ in range(int(rng.integers(0, 9))):
            out.append(items(rng) + rng.choice(nl_choices))
    data = b"".join(out)
    if rng.random() < 0.2 and data.endswith(b"\n"):
        data = data[:-1]                                         # no final newline
    if rng.random() < 0.03:
        data += b"\xff\xfe\n>after\n1\n"                         # invalid UTF-8
    return data


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    cmds = sys.argv[2:] or ["seedextend", "uniq"]
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
    work = tempfile.mkdtemp(prefix="umgap_fuzz")
    from umgap_amd.synth import synthetic_taxonomy, write_taxonomy_tsv
    ids, par, valid, rank = synthetic_taxonomy(300, 500, max_depth=8, seed=4)
    tsv = os.path.join(work, "taxons.tsv")
    write_taxonomy_tsv(tsv, ids, par, valid, rank)
    kmers = sorted({"".join(rng.choice(list(LETTERS), 9)) for _ in range(400)})
    idx_tsv = os.path.join(work, "index.tsv")
    with open(idx_tsv, "w") as f:
        for k in kmers:
            f.write("%s\t%d\n" % (k, int(rng.choice(np.concatenate([ids, [499, 100000]])))))
    idx = os.path.join(work, "index.umgapix")
    peps = sorted({"".join(rng.choice(list(LETTERS), int(rng.integers(4, 14)))) + str(rng.choice(["K", "R"])) for _ in range(300)})
    tryp_tsv = os.path.join(work, "tryptic.tsv")
    with open(tryp_tsv, "w") as f:
        for k in peps:
            f.write("%s\t%d\n" % (k, int(rng.choice(ids))))
    tryp = os.path.join(work, "tryptic.umgapix")
    need_dev = any(c in ("taxa2agg", "taxa2agg_scored", "pept2lca", "pept2lca_var", "prot2tryp2lca", "prot2kmer2lca", "prot2kmer2agg") for c in cmds)
    if need_dev:
        with open(idx_tsv, "rb") as fi, open(idx, "wb") as fo:
            subprocess.run([UMGAP, "buildindex"], stdin=fi, stdout=fo, check=True)
        with open(tryp_tsv, "rb") as fi, open(tryp, "wb") as fo:
            subprocess.run([UMGAP, "buildindex"], stdin=fi, stdout=fo, check=True)

    def taxa_item(rng):
        c = rng.random()
        if c < 0.85:
            return b"%d" % int(rng.choice(np.concatenate([ids[:20], [0, 0, 0, 499, 7]])))
        return rng.choice([b"", b"abc", b"-1", b"+5", b" 5", b"5 ", b"18446744073709551616", b"00012", b"1=2"])

    def pep_item(rng):
        c = rng.random()
        if c < 0.5:
            return kmers[int(rng.integers(0, len(kmers)))].encode()
        if c < 0.8:
            return "".join(rng.choice(list(LETTERS + "*X"), int(rng.integers(0, 30)))).encode()
        return rng.choice([b"", b"ACDEFGHIk", b"ACD\xc3\xa9FGHIKLMNP", b"ACDEFGHI ", b">x"])

    def scored_item(rng):
        c = rng.random()
        if c < 0.85:
            return b"%d=%s" % (int(rng.choice(np.concatenate([ids[:20], [0, 0, 499]]))), rng.choice([b"1", b"0.5", b"2.25", b"1e-3", b"3", b"0.1"]))
        return rng.choice([b"", b"5", b"5=", b"=1", b"5=1=2", b"5=abc", b"5=inf", b"5=-1"])

    def tryp_pep_item(rng):
        c = rng.random()
        if c < 0.6:
            return peps[int(rng.integers(0, len(peps)))].encode()
        return "".join(rng.choice(list(LETTERS + "KR*"), int(rng.integers(0, 20)))).encode()

    def tryp_prot_item(rng):
        parts = []
        for _ in range(int(rng.integers(0, 6))):
            parts.append(peps[int(rng.integers(0, len(peps)))] if rng.random() < 0.6 else "".join(rng.choice(list(LETTERS + "KRP*"), int(rng.integers(0, 15)))))
        return "".join(parts).encode()

    def prot_item(rng):
        n = int(rng.integers(0, 40))
        s = "".join(rng.choice(list(LETTERS + "*"), n))
        if n > 12 and rng.random() < 0.6:
            at = int(rng.integers(0, n - 9))
            s = s[:at] + kmers[int(rng.integers(0, len(kmers)))] + s[at + 9:]
        return s.encode()

    bad = 0
    for case in range(n_cases):
        if case % 20 == 0:
            print("case %d, %d mismatches so far" % (case, bad), flush=True)
        for cmd in cmds:
            if cmd == "seedextend":
                flags = [["-g", "1", "-s", "3"], [], ["-s", "1"], ["-g", "2", "-s", "2", "-r", tsv], ["-r", tsv, "-p", "1"],
                         ["-s", "1", "-g", "1", "-r", tsv], ["-s", "1", "-g", "2", "-r", tsv, "-p", "0"], ["-s", "0", "-g", "1", "-r", tsv, "-p", "3"]][int(rng.integers(0, 8))]
                data, mine, theirs = soup(rng, taxa_item, int(rng.integers(0, 12))), ["seedextend"] + flags, ["seedextend"] + flags
            elif cmd == "uniq":
                flags = [[], ["-d", "/"], ["-d", "/", "-s", " "], ["-w"], ["-d", "r", "-w"]][int(rng.integers(0, 5))]
                data, mine, theirs = soup(rng, taxa_item, int(rng.integers(0, 12))), ["uniq"] + flags, ["uniq"] + flags
            elif cmd == "taxa2agg":
                flags = [["-a", "lca*", "-l", "2"], [], ["-m", "rmq", "-a", "mrtl"], ["-m", "rmq", "-a", "lca*"], ["-r", "-a", "lca*"], ["-m", "rmq", "-f", "0.6"]][int(rng.integers(0, 6))]
                data, mine, theirs = soup(rng, taxa_item, int(rng.integers(0, 12))), ["taxa2agg"] + flags + [tsv], ["taxa2agg"] + flags + [tsv]
            elif cmd == "taxa2agg_scored":
                flags = [["-s"], ["-s", "-a", "lca*", "-l", "1.5"], ["-s", "-m", "rmq", "-a", "mrtl"], ["-s", "-m", "rmq", "-a", "lca*", "-l", "0.6"]][int(rng.integers(0, 4))]
                data, mine, theirs = soup(rng, scored_item, int(rng.integers(0, 12))), ["taxa2agg"] + flags + [tsv], ["taxa2agg"] + flags + [tsv]
            elif cmd == "pept2lca_var":
                flags = [[], ["-o"], ["-o", "-c", "3"]][int(rng.integers(0, 3))]
                data, mine, theirs = soup(rng, tryp_pep_item, int(rng.integers(0, 12))), ["pept2lca"] + flags + [tryp], ["pept2lca"] + flags + [tryp_tsv]
            elif cmd == "prot2tryp2lca":
                flags = [[], ["-o"], ["-l", "9", "-L", "45"], ["-o", "-l", "3", "-L", "12", "-c", "2"], ["-k", "AC"], ["-d", "W"]][int(rng.integers(0, 6))]
                data, mine, theirs = soup(rng, tryp_prot_item, int(rng.integers(0, 12))), ["prot2tryp2lca"] + flags + [tryp], ["prot2tryp2lca"] + flags + [tryp_tsv]
            elif cmd == "pept2lca":
                flags = [[], ["-o"], ["-o", "-c", "3"]][int(rng.integers(0, 3))]
                data, mine, theirs = soup(rng, pep_item, int(rng.integers(0, 12))), ["pept2lca"] + flags + [idx], ["pept2lca"] + flags + [idx_tsv]
            elif cmd == "prot2kmer2lca":
                flags = [[], ["-o"], ["-o", "-c", "2"]][int(rng.integers(0, 3))]
                data, mine, theirs = soup(rng, prot_item, int(rng.integers(0, 12))), ["prot2kmer2lca"] + flags + [idx], ["prot2kmer2lca"] + flags + [idx_tsv]
            elif cmd == "prot2kmer2agg":
                se, uq, ag = [(["-g", "1", "-s", "3"], ["-d", "/"], ["-l", "2", "-a", "lca*"]), (["-s", "2"], [], ["-a", "hybrid"]),
                              (["-g", "1", "-s", "2"], ["-d", "/"], ["-l", "1", "-m", "rmq", "-a", "mrtl"]),
                              (["-g", "1", "-s", "1"], ["-d", "/"], ["-l", "1", "-a", "lca*"]), (["-g", "2", "-s", "0"], [], ["-m", "rmq", "-a", "hybrid"])][int(rng.integers(0, 5))]
                data = soup(rng, prot_item, int(rng.integers(0, 12)))
                stage, ok = data, True
                for theirs in (["prot2kmer2lca", "-o", idx_tsv], ["seedextend"] + se, ["uniq"] + uq, ["taxa2agg"] + ag + [tsv]):
                    orc, stage, _ = O.run_cli(theirs, stage)
                    ok = ok and orc == 0
                if not ok:
                    continue          # error paths of the fused command are its own
                env = {"UMGAP_BLOCK_BYTES": str(int(rng.choice([32, 100, 1 << 20]))), "UMGAP_THREADS": str(int(rng.integers(1, 6)))}
                rc, out = run(["prot2kmer2agg"] + se + uq + ag + [idx, tsv], data, env)
                if rc != 0 or out != stage:
                    bad += 1
                    print("MISMATCH prot2kmer2agg", se, uq, ag, env, "rc", rc, "input", repr(data[:300]), "out", out[:120], "want", stage[:120])
                    if bad >= 5:
                        raise SystemExit(1)
                continue
            else:
                raise SystemExit("unknown command " + cmd)
            env = {"UMGAP_BLOCK_BYTES": str(int(rng.choice([32, 100, 1 << 20]))), "UMGAP_THREADS": str(int(rng.integers(1, 6)))}
            rc, out = run(mine, data, env)
            orc, oout, _ = O.run_cli(theirs, data)
            if rc != orc or out != oout:
                bad += 1
                path = os.path.join(work, "fail_%s_%d.bin" % (cmd, case))
                open(path, "wb").write(data)
                print("MISMATCH", cmd, mine[1:], env, "rc", rc, orc, "input", repr(data[:400]), "out", out[:120], "want", oout[:120], "stderr", run.last_stderr[-600:])
                if bad >= 5:
                    raise SystemExit(1)
    print("fuzz: %d cases x %s, %d mismatches" % (n_cases, cmds, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
