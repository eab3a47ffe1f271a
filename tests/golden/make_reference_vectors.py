#!/usr/bin/env python3
"""Extracts tests/golden/reference_vectors.json from the reference's own unit tests.

The vectors are the known-answer data held by the unit tests and doc examples of unipept/umgap
v1.1.1 for the pept2lca -> taxa2agg path.  Nothing is typed by hand: every block is pulled out of
the reference's Rust sources with regular expressions (the sources are read as text; the
reference is Rust and cannot be executed in this image), so a transcription slip cannot happen.
Only the resulting JSON -- data: inputs and expected outputs -- is committed.

    python tests/golden/make_reference_vectors.py            # rewrite the JSON
    python tests/golden/make_reference_vectors.py --check    # exit 1 if the committed JSON differs

`--check` runs in tests/test_oracle_golden.py whenever /root/reference is present (the build
container); on the GPU box only the JSON exists.
"""
import json
import os
import re
import sys

REF = os.environ.get("UMGAP_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_vectors.json")


def src(rel):
    with open(os.path.join(REF, rel), encoding="utf-8") as f:
        return f.read()


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def line_of(text, pos):
    return text.count("\n", 0, pos) + 1


def fn_body(text, name):
    """(body, first_line, last_line) of `fn name(...) {...}` by brace matching."""
    m = re.search(r"fn\s+%s\s*\([^)]*\)[^{]*\{" % re.escape(name), text)
    if not m:
        raise SystemExit("function %s not found" % name)
    depth, i = 1, m.end()
    while depth:
        c = text[i]
        depth += (c == "{") - (c == "}")
        i += 1
    return text[m.end():i - 1], line_of(text, m.start()), line_of(text, i - 1)


def ints(s):
    return [int(x) for x in re.findall(r"\d+", strip_comments(s))]


def rank_names():
    """Rank variant -> (index, serialized name), src/rank.rs:9-42."""
    text = src("src/rank.rs")
    body = text[text.index("pub enum Rank"):]
    body = body[:body.index("}")]
    out = {}
    for i, m in enumerate(re.finditer(r'serialize\s*=\s*"([^"]+)"\)\]\s*(\w+)', body)):
        out[m.group(2)] = (i, m.group(1))
    assert len(out) == 32, len(out)
    return out


RANKS = rank_names()


def taxa_from_static(body):
    rows = []
    for m in re.finditer(r'Taxon::from_static\(\s*(\d+)\s*,\s*"([^"]*)"\s*,\s*Rank::(\w+)\s*,\s*(\d+)\s*,\s*(true|false)\s*\)', body):
        rows.append([int(m.group(1)), m.group(2), RANKS[m.group(3)][1], int(m.group(4)), m.group(5) == "true"])
    return rows


def rust_str(lit):
    """Decodes the body of a Rust string literal as far as the test sources use escapes."""
    return re.sub(r"\\x([0-9a-fA-F]{2})", lambda m: chr(int(m.group(1), 16)), lit).replace("\\t", "\t").replace("\\n", "\n")


def counting_cases(body):
    """assert_matches!(agg.counting_aggregate(&vec![..]), Ok(x)) and the assert!(vec![..].contains(..)) tie-set form,
    in source order, as (input, [accepted outcomes])."""
    cases = []
    pat = re.compile(r"assert_matches!\(\s*\w+\.counting_aggregate\(&vec!\[([^\]]*)\]\)\s*,\s*Ok\((\d+)\)\s*\)"
                     r"|assert!\(\s*vec!\[([^\]]*)\]\.contains\(&\w+\.counting_aggregate\(&vec!\[([^\]]*)\]\)\.unwrap\(\)\)\)")
    for m in pat.finditer(body):
        if m.group(1) is not None:
            cases.append((ints(m.group(1)), [int(m.group(2))]))
        else:
            cases.append((ints(m.group(4)), ints(m.group(3))))
    return cases


def dedup(cases):
    seen, out = set(), []
    for c in cases:
        k = json.dumps(c)
        if k not in seen:
            seen.add(k)
            out.append(c)
    return out


def doc_examples(rel, struct):
    """The ```sh blocks of the doc comment above `pub struct <struct>`: list of (command line, lines) sections."""
    text = src(rel)
    head = text[:text.index("pub struct " + struct)]
    doc = "\n".join(re.sub(r"^\s*/// ?", "", l) for l in head.split("\n") if l.strip().startswith("///"))
    blocks = []
    for b in re.findall(r"```sh\n(.*?)```", doc, flags=re.S):
        sections = []
        for l in b.rstrip("\n").split("\n"):
            if l.startswith("$ "):
                sections.append((l[2:], []))
            else:
                sections[-1][1].append(l)
        blocks.append(sections)
    return blocks, text


def build():
    g = {"_doc": "Golden vectors held by the reference's own unit tests and doc examples for the pept2lca->taxa2agg path "
                 "(unipept/umgap v1.1.1), EXTRACTED from its sources by tests/golden/make_reference_vectors.py. "
                 "Data only; each block cites the reference file:line it was read from."}

    # ---- fixtures --------------------------------------------------------------------------------
    t = src("src/fixtures.rs")
    body, a, b = fn_body(t, "taxon_list")
    root = int(re.search(r"pub const ROOT: TaxonId = (\d+);", t).group(1))
    g["fixture_taxonomy"] = {"source": "src/fixtures.rs:%d-%d" % (a, b), "root": root, "rows": taxa_from_static(body)}
    assert len(g["fixture_taxonomy"]["rows"]) == 6

    t = src("src/rmq/lca.rs")
    body, a, b = fn_body(t, "large_taxon_list")
    g["large_taxonomy"] = {"source": "src/rmq/lca.rs:%d-%d" % (a, b),
                           "rows": [[int(x), int(y)] for x, y in re.findall(r"taxon\((\d+),\s*(\d+)\)", body)]}

    # ---- taxon parsing ---------------------------------------------------------------------------
    t = src("src/taxon.rs")
    body, a, b = fn_body(t, "test_taxon_parsing")
    ok, err = [], []
    for m in re.finditer(r'assert_eq!\(Taxon::from_static\(\s*(\d+)\s*,\s*"([^"]*)"\s*,\s*Rank::(\w+)\s*,\s*(\d+)\s*,\s*(true|false)\s*\)\s*,\s*"([^"]*)"\.parse\(\)', body):
        ok.append([rust_str(m.group(6)), [int(m.group(1)), m.group(2), RANKS[m.group(3)][0], int(m.group(4)), m.group(5) == "true"]])
    for m in re.finditer(r'assert_matches!\(\*"([^"]*)"\.parse::<Taxon>\(\)\.unwrap_err\(\)\.kind\(\)\s*,\s*ErrorKind::(\w+)', body):
        err.append([rust_str(m.group(1)), m.group(2)])
    assert len(ok) == 4 and len(err) == 6, (len(ok), len(err))
    # the doctest of FromStr (a `let taxon = "…".parse::<Taxon>()` followed by the struct it returns, in comments)
    dm = re.search(r'///\s*let taxon = "([^"]*)"\.parse::<Taxon>\(\);(.*?)///\s*```', t, flags=re.S)
    fields = dict(re.findall(r"//\s+(\w+):\s*([^,\n]*)", dm.group(2)))
    dline = line_of(t, dm.start())
    g["taxon_parsing"] = {"source": "src/taxon.rs:%d-%d" % (a, b), "ok": ok, "err": err,
                          "doctest": {"source": "src/taxon.rs:%d-%d" % (dline, line_of(t, dm.end())), "line": rust_str(dm.group(1)),
                                      "want": [int(fields["id"]), fields["name"].strip('"'), RANKS[fields["rank"].split("::")[-1]][0],
                                               int(fields["parent"]), fields["valid"].strip() == "true"]}}

    body, a, b = fn_body(t, "test_euler_tour")
    g["euler_tour"] = {"source": "src/taxon.rs:%d-%d" % (a, b),
                       "want": [[int(x), int(y)] for x, y in re.findall(r"\((\d+),\s*(\d+)\)", body[body.index("vec!["):])]}
    body, a, b = fn_body(t, "test_taxon_list")
    anc = {}
    for m in re.finditer(r"assert_eq!\((Some\((\d+)\)|None),\s*ancestry\[(\d+)\]\)", body):
        anc[m.group(3)] = int(m.group(2)) if m.group(2) else None
    g["taxon_list"] = {"source": "src/taxon.rs:%d-%d" % (a, b), "ancestry": anc}
    body, a, b = fn_body(t, "test_lineage")
    rows = g["fixture_taxonomy"]["rows"]
    lin = {"source": "src/taxon.rs:%d-%d" % (a, b), "of": int(re.search(r"lineage\((\d+)\)", body).group(1))}
    for m in re.finditer(r"rank == Rank::(\w+)\s*\{\s*assert_eq!\(lineage\[rank\]\.as_ref\(\),\s*Some\(&list\[(\d+)\]\)\)", body):
        lin[RANKS[m.group(1)][1]] = rows[int(m.group(2))][0]
    g["lineage"] = lin

    # ---- RMQ -------------------------------------------------------------------------------------
    t = src("src/rmq/mod.rs")
    first = line_of(t, t.index("fn test_block_minima"))
    body, _, _ = fn_body(t, "test_block_minima")
    m = re.search(r"else\s*\{\s*vec!\[([^\]]*)\]\s*\}\s*,\s*RMQ::block_min\(&\[([^\]]*)\]\)", body)
    block_min = {"array": ints(m.group(2)), "want64": ints(m.group(1))}
    body, _, _ = fn_body(t, "test_sparse")
    m = re.search(r"assert_eq!\(\s*vec!\[(.*?)\]\s*,\s*RMQ::sparse\(\s*&\[([^\]]*)\]\s*,\s*&\[([^\]]*)\]", body, flags=re.S)
    sparse = {"array": ints(m.group(2)), "block_min": ints(m.group(3)), "want": [ints(x) for x in re.findall(r"vec!\[([^\]]*)\]", m.group(1))]}
    body, _, _ = fn_body(t, "array")
    array200 = ints(body[body.index("vec!["):])
    assert len(array200) == 200
    queries = []
    for name in ("test_rmq_single_block", "test_rmq_two_blocks", "test_rmq_three_blocks", "test_rmq_more_blocks"):
        body, _, _ = fn_body(t, name)
        queries += [[int(x), int(y), int(w)] for w, x, y in re.findall(r"assert_eq!\((\d+),\s*info\.query\((\d+),\s*(\d+)\)\)", body)]
    waves = {}
    for name in ("test_wave_of_33", "test_wave_of_65"):
        body, _, last = fn_body(t, name)
        arr = ints(re.search(r"let array = vec!\[([^\]]*)\]", body).group(1))
        w, x, y = re.search(r"assert_eq!\((\d+),\s*info\.query\((\d+),\s*(\d+)\)\)", body).groups()
        waves[name[-2:]] = {"array": arr, "query": [int(x), int(y), int(w)]}
    g["rmq"] = {"source": "src/rmq/mod.rs:%d-%d" % (first, last), "block_min": block_min, "sparse": sparse, "array200": array200,
                "queries": queries, "wave33": waves["33"], "wave65": waves["65"]}

    # ---- aggregators -----------------------------------------------------------------------------
    def cases_of(rel, names):
        t = src(rel)
        out, lo, hi = [], None, None
        for n in names:
            body, a, b = fn_body(t, n)
            out += counting_cases(body)
            lo, hi = a if lo is None else min(lo, a), b if hi is None else max(hi, b)
        return out, "%s:%d-%d" % (rel, lo, hi)

    tri = ("test_two_on_same_path", "test_two_on_fork", "test_three_on_triangle")
    c_rmq, s_rmq = cases_of("src/rmq/lca.rs", tri)
    c_tree, s_tree = cases_of("src/tree/lca.rs", tri)
    assert c_rmq == c_tree, "the rmq and tree LCA* tests are expected to hold the same cases"
    for inp, want in c_rmq:
        assert len(want) == 1
    g["lca_tests"] = {"source": s_rmq + ", " + s_tree, "cases": [[i, w[0]] for i, w in c_rmq]}
    c, s = cases_of("src/rmq/lca.rs", ("test_with_deeper_interns",))
    g["lca_deeper_interns"] = {"source": s, "cases": dedup([[i, w[0]] for i, w in c])}
    c, s = cases_of("src/rmq/rtl.rs", ("test_all_on_same_path", "favouring_root", "leaning_close", "non_deterministic"))
    g["mrtl_tests"] = {"source": s, "cases": [[i, w] for i, w in c]}

    def mix_cases(rel, names):
        t = src(rel)
        out, lo, hi = [], None, None
        for n in names:
            body, a, b = fn_body(t, n)
            f = float(re.search(r"MixCalculator::new\([^;]*?,\s*([0-9.]+)\s*\);", body).group(1))
            out += [[f, i, w] for i, w in counting_cases(body)]
            lo, hi = a if lo is None else min(lo, a), b if hi is None else max(hi, b)
        return dedup(out), "%s:%d-%d" % (rel, lo, hi)

    c, s = mix_cases("src/tree/mix.rs", ("test_full_rtl", "test_full_lca", "test_two_thirds"))
    g["tree_mix_tests"] = {"source": s, "cases": c}
    c, s = mix_cases("src/rmq/mix.rs", ("test_full_rtl", "test_full_lca", "test_one_half"))
    g["rmq_mix_tests"] = {"source": s, "cases": c}

    t = src("src/agg/mod.rs")
    body, a, _ = fn_body(t, "aggregators")
    aggs = []
    for m in re.finditer(r"Box::new\((rmq|tree)::(lca|rtl|mix)::\w+::new\(([^;]*?)\)\)\s*,", body):
        strat = {"lca": "lca*", "rtl": "mrtl", "mix": "hybrid"}[m.group(2)]
        f = re.findall(r"([0-9]+\.[0-9]+)", m.group(3))
        aggs.append([m.group(1), strat, float(f[0]) if f else 0.0])
    assert len(aggs) == 9
    body, _, b = fn_body(t, "test_invalid_taxa")
    g["agg_contract"] = {"source": "src/agg/mod.rs:%d-%d" % (a, b), "aggregators": aggs,
                         "unknown_inputs": [ints(x) for x in re.findall(r"counting_aggregate\(&vec!\[([^\]]*)\]\)", body)],
                         "unknown_payload": int(re.search(r"UnknownTaxon\((\d+)\)", body).group(1))}

    # ---- documentation examples (text level) -----------------------------------------------------
    def fasta(lines, spaces_to_newlines=False):
        out = []
        for l in lines:
            if spaces_to_newlines and not l.startswith(">"):
                out += l.split(" ")
            else:
                out.append(l)
        return "".join(x + "\n" for x in out)

    def where(text, struct, n_lines_hint=None):
        end = line_of(text, text.index("pub struct " + struct)) - 1
        head = text[:text.index("pub struct " + struct)].split("\n")
        i = len(head) - 2
        while i >= 0 and head[i].strip().startswith("///"):
            i -= 1
        return "%d-%d" % (i + 2, end)

    ex = {}
    blocks, text = doc_examples("src/commands/uniq.rs", "Uniq")
    (c0, l0), (c1, l1) = blocks[0]
    assert c0.startswith("cat ") and c1.startswith("umgap uniq ")
    ex["uniq"] = {"source": "src/commands/uniq.rs:" + where(text, "Uniq"), "args": c1.split()[2:c1.split().index("<")],
                  "stdin": fasta(l0), "stdout": fasta(l1)}
    blocks, text = doc_examples("src/commands/seedextend.rs", "SeedExtend")
    secs = blocks[0]
    inp = [s for s in secs if "prot2kmer2lca" in s[0]][0][1]
    outp = [s for s in secs if s[0].startswith("umgap seedextend")][0][1]
    ex["seedextend"] = {"source": "src/commands/seedextend.rs:" + where(text, "SeedExtend"), "args": [],
                        "stdin": fasta(inp, True), "stdout": fasta(outp, True)}
    blocks, text = doc_examples("src/commands/pept2lca.rs", "PeptToLca")
    cases, index = [], {}
    for secs in blocks:
        (c0, l0), (c1, l1) = secs
        args = [a for a in c1.split()[2:c1.split().index("<")] if a.startswith("-")]
        cases.append({"args": args, "stdin": fasta(l0), "stdout": fasta(l1)})
        for pep, tax in zip(l0[1:], l1[1:]):
            if tax != "0":
                index[pep] = int(tax)
    ex["pept2lca"] = {"source": "src/commands/pept2lca.rs:" + where(text, "PeptToLca"),
                      "index_tsv": "".join("%s\t%d\n" % kv for kv in sorted(index.items())), "cases": cases}
    blocks, text = doc_examples("src/commands/prot2kmer2lca.rs", "ProtToKmerToLca")
    (c0, l0), (c1, l1) = blocks[0]
    k = int(re.search(r"-k\s*(\d+)", c1).group(1))
    seq = l0[1]
    ex["prot2kmer2lca_shape"] = {"source": "src/commands/prot2kmer2lca.rs:" + where(text, "ProtToKmerToLca") +
                                           " (the values need the real index; shape only)",
                                 "stdin": fasta(l0), "k": k, "windows": len(seq) - k + 1,
                                 "hits_without_o": len(l1) - 1, "lines_with_o": len(blocks[1][0][1]) - 1}
    g["text_examples"] = ex
    return g


def testdata():
    """All read pairs of the reference's manual test data (testdata/A1.fq, A2.fq; testdata/README.md): BASELINE.json configs[0]."""
    out = {"_doc": "The 100 read pairs of the reference's manual test data testdata/A1.fq / A2.fq (100-bp paired reads, taxon id "
                   "embedded in the header, testdata/README.md). Data only; written by tests/golden/make_reference_vectors.py."}
    for name in ("A1", "A2"):
        lines = src("testdata/%s.fq" % name).split("\n")
        recs = []
        for i in range(0, len(lines) - 3, 4):
            assert lines[i].startswith("@") and lines[i + 2].startswith("+")
            recs.append({"header": lines[i][1:], "seq": lines[i + 1], "qual": lines[i + 3]})
        out[name] = recs
    assert len(out["A1"]) == len(out["A2"]) == 100
    return out


OUT_TESTDATA = os.path.join(HERE, "testdata_A.json")


def main():
    docs = ((OUT, build()), (OUT_TESTDATA, testdata()))
    if "--check" in sys.argv:
        for path, g in docs:
            old = open(path).read() if os.path.exists(path) else ""
            if json.loads(old or "{}") != g:
                sys.stderr.write("%s differs from what the reference's sources hold; rerun tests/golden/make_reference_vectors.py\n" % path)
                return 1
        print("the committed golden files match the reference's sources")
        return 0
    for path, g in docs:
        with open(path, "w") as f:
            f.write(json.dumps(g, indent=1) + "\n")
        print("wrote", path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
