"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
inputs.  Integer/index work => bit-exact.  Run on the MI355X box: `pytest -m gpu`."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


@pytest.fixture(scope="module")
def api(torch_cuda):
    from umgap_amd import api
    return api


def _golden():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


def _fixture_taxonomy(api):
    from umgap_amd.api import RANKS
    rows = _golden()["fixture_taxonomy"]["rows"]
    ids = [r[0] for r in rows]
    par = [r[3] for r in rows]
    valid = [int(r[4]) for r in rows]
    rank = [RANKS.index(r[2]) for r in rows]
    return api.Taxonomy(ids, par, valid, rank), O.Taxonomy.from_arrays(ids, par, valid, rank)


def _csr(reads):
    offs = np.zeros(len(reads) + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    taxa = np.array([t for r in reads for t in r], np.uint32)
    return taxa, offs


# ---------------------------------------------------------------------------------------------
# index: get() semantics (the fst seam: get(k) = v iff (k, v) was inserted)
# ---------------------------------------------------------------------------------------------
def test_lookup_matches_dict(api):
    rng = np.random.default_rng(1)
    n = 300_000
    keys = np.unique(rng.integers(0, 1 << 45, n, dtype=np.uint64))
    vals = rng.integers(1, 1 << 32, len(keys), dtype=np.uint64).astype(np.uint32)
    ix = api.Index.build(keys, vals)
    assert ix.num_keys == len(keys)
    # every key present
    assert (ix.lookup9(keys) == vals).all()
    # mixed present / absent / out-of-range, ragged length (not a multiple of 64)
    q = np.concatenate([keys[:5001], rng.integers(0, 1 << 45, 7777, dtype=np.uint64), np.array([1 << 45, (1 << 64) - 1], np.uint64)])
    rng.shuffle(q)
    d = dict(zip(keys.tolist(), vals.tolist()))
    want = np.array([d.get(int(k), 0) for k in q], np.uint32)
    assert (ix.lookup9(q, miss_value=0) == want).all()
    want7 = np.array([d.get(int(k), 7) for k in q], np.uint32)
    assert (ix.lookup9(q, miss_value=7) == want7).all()
    # empty batch
    assert len(ix.lookup9(np.zeros(0, np.uint64))) == 0


def test_lookup_overflow_chains(api):
    """A table filled to 95 % forces displacement chains; results must not change."""
    rng = np.random.default_rng(2)
    keys = np.unique(rng.integers(0, 1 << 45, 900_000, dtype=np.uint64))
    vals = (np.arange(len(keys)) % 1000 + 1).astype(np.uint32)
    for load in (0.3, 0.8, 0.95):
        ix = api.Index.build(keys, vals, load_factor=load)
        assert (ix.lookup9(keys) == vals).all(), load
        absent = rng.integers(0, 1 << 45, 100_000, dtype=np.uint64)
        absent = absent[~np.isin(absent, keys)]
        assert (ix.lookup9(absent, miss_value=0) == 0).all(), load


def test_lookup_variants_and_device_path(api, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(3)
    keys = np.unique(rng.integers(0, 1 << 45, 200_000, dtype=np.uint64))
    vals = rng.integers(1, 1 << 31, len(keys)).astype(np.uint32)
    ix = api.Index.build(keys, vals, load_factor=0.85)
    q = np.concatenate([keys[::3], rng.integers(0, 1 << 45, 50_001, dtype=np.uint64)])
    want = ix.lookup9(q)
    dq = torch.from_numpy(q.view(np.int64)).cuda()
    for variant in ("quad", "lane"):
        got = ix.lookup9_dev(dq, variant=variant)
        torch.cuda.synchronize()
        assert (got.cpu().numpy().view(np.uint32) == want).all(), variant


def test_duplicate_keys_are_refused(api):
    keys = np.array([5, 6, 5], np.uint64)
    with pytest.raises(api.UmgapError, match="duplicate"):
        api.Index.build(keys, np.array([1, 2, 3], np.uint32))
    with pytest.raises(api.UmgapError, match="45-bit"):
        api.Index.build(np.array([1 << 45], np.uint64), np.array([1], np.uint32))


def test_index_file_roundtrip_and_shards(api, tmp_path):
    rng = np.random.default_rng(4)
    keys = np.unique(rng.integers(0, 1 << 45, 400_000, dtype=np.uint64))
    vals = rng.integers(1, 1 << 31, len(keys)).astype(np.uint32)
    ix = api.Index.build(keys, vals, load_factor=0.9)
    path = tmp_path / "nine.umgapix"
    ix.save(path)
    q = np.concatenate([keys, rng.integers(0, 1 << 45, 10_000, dtype=np.uint64)])
    want = ix.lookup9(q)
    ix2 = api.Index.open(path)
    assert ix2.num_keys == len(keys) and (ix2.lookup9(q) == want).all()
    # shards: every key is answered by exactly one shard, misses by none
    for n_shards in (2, 3, 8):
        acc = np.zeros(len(q), np.uint32)
        hits = np.zeros(len(q), np.int32)
        for s in range(n_shards):
            sh = api.Index.open(path, shard=s, n_shards=n_shards)
            r = sh.lookup9(q, miss_value=0)
            acc |= r
            hits += (r != 0)
        assert (acc == want).all() and hits.max() == 1
    with pytest.raises(api.UmgapError):
        api.Index.open(tmp_path / "missing.umgapix")
    bad = tmp_path / "bad.fst"
    bad.write_bytes(b"\x02" + b"\x00" * 63)
    with pytest.raises(api.UmgapError, match="UMGAPIX1"):
        api.Index.open(bad)


def test_pept2lca_doc_example(api):
    """src/commands/pept2lca.rs:24-46 on a 9-mer index (the doc's peptides are tryptic; shape only)."""
    keys = [api.pack9(b"AAALTERAA"), api.pack9(b"ENFVYLAKW")]
    ix = api.Index.build(np.array(keys, np.uint64), np.array([2, 3398], np.uint32))
    q = np.array([api.pack9(b"AAALTERAA"), api.pack9(b"NOTATRYPT"), api.pack9(b"ENFVYLAKW")], np.uint64)
    assert ix.lookup9(q, miss_value=0).tolist() == [2, 0, 3398]


# ---------------------------------------------------------------------------------------------
# aggregate: the reference's own known-answer tests, through the C ABI
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("method", ["tree", "rmq"])
def test_lca_star_golden(api, method):
    tx, _ = _fixture_taxonomy(api)
    g = _golden()
    reads = [c[0] for c in g["lca_tests"]["cases"]]
    want = [c[1] for c in g["lca_tests"]["cases"]]
    taxa, offs = _csr(reads)
    t, s = tx.aggregate(taxa, offs, method=method, strategy="lca*")
    assert s.tolist() == [0] * len(reads) and t.tolist() == want


def test_lca_star_deeper_interns(api):
    g = _golden()
    rows = g["large_taxonomy"]["rows"]
    tx = api.Taxonomy([r[0] for r in rows], [r[1] for r in rows])
    reads = [c[0] for c in g["lca_deeper_interns"]["cases"]]
    taxa, offs = _csr(reads)
    for method in ("tree", "rmq"):
        t, s = tx.aggregate(taxa, offs, method=method, strategy="lca*")
        assert t.tolist() == [c[1] for c in g["lca_deeper_interns"]["cases"]] and not s.any()


@pytest.mark.parametrize("method", ["tree", "rmq"])
def test_agg_contract_lca(api, method):
    """src/agg/mod.rs:67-118 at the record level: [] -> "1"/EMPTY, [t] -> t, unknown 5 -> UnknownTaxon(5)."""
    tx, _ = _fixture_taxonomy(api)
    ids = [1, 2, 10239, 12884, 185751, 185752]
    reads = [[]] + [[t] for t in ids] + [[5], [1, 2, 5, 1], [0, 0]]
    taxa, offs = _csr(reads)
    t, s = tx.aggregate(taxa, offs, method=method, strategy="lca*")
    assert (t[0], s[0]) == (1, api.READ_EMPTY)
    assert t[1:7].tolist() == ids and not s[1:7].any()
    assert (t[7], s[7]) == (5, api.READ_UNKNOWN_TAXON) and (t[8], s[8]) == (5, api.READ_UNKNOWN_TAXON)
    assert (t[9], s[9]) == (1, api.READ_EMPTY)
    # id beyond the list: tree panics (index out of bounds), rmq answers UnknownTaxon (HashMap miss)
    t, s = tx.aggregate(np.array([999999], np.uint32), np.array([0, 1], np.uint64), method=method, strategy="lca*")
    assert s[0] == (api.READ_PANIC if method == "tree" else api.READ_UNKNOWN_TAXON)


def test_tree_mrtl_is_an_invalid_invocation(api):
    tx, _ = _fixture_taxonomy(api)
    with pytest.raises(api.UmgapError, match="cannot be combined"):
        tx.aggregate(np.array([1], np.uint32), np.array([0, 1], np.uint64), method="tree", strategy="mrtl")


def test_taxonomy_tables_match_oracle(api):
    from umgap_amd.synth import synthetic_taxonomy
    ids, par, valid, rank = synthetic_taxonomy(30_000, 50_000, max_depth=40)
    tx = api.Taxonomy(ids, par, valid, rank)
    ot = O.Taxonomy.from_arrays(ids, par, valid, rank)
    assert tx.root == ot.root and len(tx) == len(ot) and tx.reachable == len(ids)
    for ranked in (False, True):
        want = ot.snapping(ranked)
        got = tx.snapping(ranked).astype(np.uint64)
        got[got == 0xFFFFFFFF] = O.NONE
        assert (got == want).all()
    # preorder = order of first occurrence in the reference's Euler tour (src/taxon.rs:309-392)
    pre, last, depth = tx.preorder()
    seen, k = {}, 0
    for t, d in ot.euler():
        if t not in seen:
            seen[t] = (k, d)
            k += 1
    for t in ids[:2000]:
        assert seen[int(t)] == (pre[t], depth[t])


def _random_reads(rng, pool, n_reads, max_len, zero_frac=0.3):
    reads = []
    for _ in range(n_reads):
        n = int(rng.integers(0, max_len + 1))
        r = rng.choice(pool, n)
        r[rng.random(n) < zero_frac] = 0
        reads.append(r.tolist())
    return reads


@pytest.mark.parametrize("method", ["tree", "rmq"])
@pytest.mark.parametrize("lower_bound", [0.0, 1.0, 2.0, 3.0])
@pytest.mark.parametrize("group", [None, "1", "0"])
def test_lca_star_random_vs_oracle(api, method, lower_bound, group, monkeypatch):
    """group: UMGAP_AGG_LCA_GROUP forces the CSR reads through the group kernel of aggregate_group.hip (what short reads take by
    themselves; the reads of more than 128 hits here come back to the per-read kernel through its list) or keeps them off it."""
    from umgap_amd.synth import synthetic_taxonomy
    if group is not None:
        monkeypatch.setenv("UMGAP_AGG_LCA_GROUP", group)
    rng = np.random.default_rng(int(lower_bound) * 7 + (method == "rmq"))
    ids, par, valid, rank = synthetic_taxonomy(3000, 5000, max_depth=25, seed=9)
    tx = api.Taxonomy(ids, par, valid, rank)
    ot = O.Taxonomy.from_arrays(ids, par, valid, rank)
    agg = O.Aggregator(ot, method, "lca*")
    # reads drawn from small pools so that taxa repeat; lengths 0..200 cover 1, 2 and 4 chunks of 64
    reads = []
    for _ in range(60):
        pool = rng.choice(ids, int(rng.integers(1, 12)))
        reads += _random_reads(rng, pool, 50, int(rng.choice([3, 20, 84, 200])))
    taxa, offs = _csr(reads)
    for ranked in (False, True):
        t, s = tx.aggregate(taxa, offs, method=method, strategy="lca*", lower_bound=lower_bound, ranked_only=ranked)
        nbad, bad, wt, ws = agg.taxa2agg_check(taxa, offs, t, s, ranked=ranked, lower_bound=lower_bound)
        assert nbad == 0, (bad, [(reads[int(b)], t[int(b)], s[int(b)], wt[int(b)], ws[int(b)]) for b in bad[:3]])


@pytest.mark.parametrize("method", ["tree", "rmq"])
@pytest.mark.parametrize("group", [None, "0"])
def test_lca_star_unknown_and_orphans_vs_oracle(api, method, group, monkeypatch):
    """Absent ids, ids beyond the list and taxa whose parent chain is broken.  (Reads of six hits: the group kernel by default;
    group "0" keeps the per-read kernel.)"""
    if group is not None:
        monkeypatch.setenv("UMGAP_AGG_LCA_GROUP", group)
    rng = np.random.default_rng(21)
    ids = np.array([1, 2, 3, 4, 5, 6, 7, 8, 20, 21, 22, 30], np.uint32)
    #            root              orphan chain: 20 -> 21 -> 99 (absent)   22 -> 400 (beyond the list)
    par = np.array([1, 1, 1, 2, 2, 3, 3, 4, 21, 99, 400, 1], np.uint32)
    tx = api.Taxonomy(ids, par)
    ot = O.Taxonomy.from_arrays(ids, par)
    agg = O.Aggregator(ot, method, "lca*")
    pool = np.array([1, 2, 3, 4, 5, 6, 7, 8, 20, 21, 22, 30, 9, 10, 50, 99, 100, 400, 5000], np.uint32)
    reads = _random_reads(rng, pool, 3000, 6, zero_frac=0.1)
    taxa, offs = _csr(reads)
    for lb in (0.0, 2.0):
        t, s = tx.aggregate(taxa, offs, method=method, strategy="lca*", lower_bound=lb)
        nbad, bad, wt, ws = agg.taxa2agg_check(taxa, offs, t, s, lower_bound=lb)
        assert nbad == 0, [(reads[int(b)], t[int(b)], s[int(b)], wt[int(b)], ws[int(b)]) for b in bad[:5]]
        assert (s == api.READ_UNKNOWN_TAXON).any() and (s == api.READ_OK).any()


def test_lca_star_scored_vs_oracle(api):
    from umgap_amd.synth import synthetic_taxonomy
    rng = np.random.default_rng(33)
    ids, par, valid, rank = synthetic_taxonomy(2000, 3000, max_depth=20, seed=5)
    tx = api.Taxonomy(ids, par, valid, rank)
    agg = O.Aggregator(O.Taxonomy.from_arrays(ids, par, valid, rank), "tree", "lca*")
    reads = []
    for _ in range(40):
        pool = rng.choice(ids, 6)
        reads += _random_reads(rng, pool, 40, 30)
    taxa, offs = _csr(reads)
    scores = rng.random(len(taxa)).astype(np.float32)
    for lb in (0.0, 0.7, 1.5):
        t, s = tx.aggregate(taxa, offs, method="tree", strategy="lca*", lower_bound=lb, scores=scores)
        nbad, bad, _, _ = agg.taxa2agg_check(taxa, offs, t, s, scores=scores, lower_bound=lb)
        assert nbad == 0, bad


# ---------------------------------------------------------------------------------------------
# MRTL / hybrid (tree) / hybrid (rmq): golden vectors and random reads against the oracle's tie sets
# ---------------------------------------------------------------------------------------------
COMBOS = [("rmq", "mrtl", 0.0), ("tree", "hybrid", 0.25), ("tree", "hybrid", 0.0), ("tree", "hybrid", 1.0),
          ("tree", "hybrid", 0.66), ("rmq", "hybrid", 0.5), ("rmq", "hybrid", 0.0), ("rmq", "hybrid", 1.0)]


def _agg_one(tx, reads, method, strategy, factor=0.25, lower_bound=0.0, scores=None):
    taxa, offs = _csr(reads)
    return tx.aggregate(taxa, offs, method=method, strategy=strategy, factor=factor, lower_bound=lower_bound, scores=scores)


def test_mrtl_golden(api):
    tx, _ = _fixture_taxonomy(api)
    cases = _golden()["mrtl_tests"]["cases"]
    t, s = _agg_one(tx, [c[0] for c in cases], "rmq", "mrtl")
    assert not s.any()
    for got, (_, want) in zip(t.tolist(), cases):
        assert got in want
    assert t[-1] == 185751  # tie {185751, 185752}: smallest preorder index


def test_tree_mix_golden(api):
    tx, _ = _fixture_taxonomy(api)
    for factor, taxa, want in _golden()["tree_mix_tests"]["cases"]:
        t, s = _agg_one(tx, [taxa], "tree", "hybrid", factor)
        assert s[0] == 0 and t[0] in want, (factor, taxa, t[0])


def test_rmq_mix_golden(api):
    tx, _ = _fixture_taxonomy(api)
    for factor, taxa, want in _golden()["rmq_mix_tests"]["cases"]:
        t, s = _agg_one(tx, [taxa], "rmq", "hybrid", factor)
        assert s[0] == 0 and t[0] in want, (factor, taxa, t[0])


def test_agg_contract_all_aggregators(api):
    """src/agg/mod.rs:67-118 for the 9 aggregator instances (LCA* ones are covered above too)."""
    tx, _ = _fixture_taxonomy(api)
    g = _golden()["agg_contract"]
    ids = [1, 2, 10239, 12884, 185751, 185752]
    for method, strategy, factor in g["aggregators"]:
        reads = [[]] + [[t] for t in ids] + g["unknown_inputs"]
        t, s = _agg_one(tx, reads, method, strategy, factor)
        assert (t[0], s[0]) == (1, api.READ_EMPTY)
        assert t[1:7].tolist() == ids and not s[1:7].any(), (method, strategy)
        for k in (7, 8):
            assert (t[k], s[k]) == (g["unknown_payload"], api.READ_UNKNOWN_TAXON), (method, strategy)


@pytest.mark.parametrize("method,strategy,factor", COMBOS)
def test_general_random_vs_oracle(api, method, strategy, factor):
    from umgap_amd.synth import synthetic_taxonomy
    rng = np.random.default_rng(abs(hash((method, strategy, factor))) % 1000)
    ids, par, valid, rank = synthetic_taxonomy(3000, 5000, max_depth=25, seed=13)
    tx = api.Taxonomy(ids, par, valid, rank)
    agg = O.Aggregator(O.Taxonomy.from_arrays(ids, par, valid, rank), method, strategy, factor)
    reads = []
    for _ in range(60):
        pool = rng.choice(ids, int(rng.integers(1, 14)))
        reads += _random_reads(rng, pool, 40, int(rng.choice([3, 20, 84, 200])))
    # reads with many distinct taxa: > 64 distinct goes through the deferred large-scratch pass
    for k in (70, 150, 400):
        reads.append(rng.choice(ids, k).tolist())
        reads.append(np.concatenate([rng.choice(ids, k), rng.choice(ids[:50], 300)]).tolist())
    taxa, offs = _csr(reads)
    for lb in (0.0, 1.0, 2.0):
        for ranked in (False, True):
            t, s = tx.aggregate(taxa, offs, method=method, strategy=strategy, factor=factor, lower_bound=lb, ranked_only=ranked)
            nbad, bad, wt, ws = agg.taxa2agg_check(taxa, offs, t, s, ranked=ranked, lower_bound=lb)
            assert nbad == 0, (lb, ranked, [(reads[int(b)][:12], int(t[int(b)]), int(s[int(b)]), int(wt[int(b)]), int(ws[int(b)])) for b in bad[:3]])


@pytest.mark.parametrize("method,strategy,factor", [("rmq", "mrtl", 0.0), ("tree", "hybrid", 0.25), ("rmq", "hybrid", 0.5)])
def test_general_more_than_1024_distinct_taxa(api, method, strategy, factor):
    """The reference has no limit on the distinct taxa of a read (src/tree/mix.rs:43-64, src/rmq/rtl.rs:39-57): beyond the
    1024 of the large LDS scratch a third pass keeps the read's working set in global memory (up to 32 768 distinct)."""
    from umgap_amd.synth import synthetic_taxonomy
    rng = np.random.default_rng(71)
    ids, par, valid, rank = synthetic_taxonomy(8000, 12000, max_depth=30, seed=17)
    tx = api.Taxonomy(ids, par, valid, rank)
    agg = O.Aggregator(O.Taxonomy.from_arrays(ids, par, valid, rank), method, strategy, factor)
    reads = [rng.choice(ids, 5).tolist()]
    for k in (1100, 2600):
        reads.append(rng.choice(ids, k, replace=False).tolist())                                        # k distinct, once each
        reads.append(np.concatenate([rng.choice(ids, k, replace=False), rng.choice(ids[:40], 500)]).tolist())   # + a heavy core
    reads.append(rng.choice(ids, 7).tolist())
    taxa, offs = _csr(reads)
    for lb in (0.0, 2.0):
        t, s = tx.aggregate(taxa, offs, method=method, strategy=strategy, factor=factor, lower_bound=lb)
        assert not (s == 4).any()                                                                       # nothing UNSUPPORTED
        nbad, bad, wt, ws = agg.taxa2agg_check(taxa, offs, t, s, lower_bound=lb)
        assert nbad == 0, (lb, [(len(reads[int(b)]), int(t[int(b)]), int(s[int(b)]), int(wt[int(b)]), int(ws[int(b)])) for b in bad[:3]])
    # the same reads twice in one call and again afterwards: the global scratch is handed back clean
    t2, s2 = tx.aggregate(np.concatenate([taxa, taxa]), np.concatenate([offs, offs[1:] + offs[-1]]), method=method, strategy=strategy, factor=factor, lower_bound=2.0)
    n = len(reads)
    assert (t2[:n] == t).all() and (t2[n:] == t).all() and (s2[:n] == s).all() and (s2[n:] == s).all()


@pytest.mark.parametrize("method,strategy,factor", [("rmq", "mrtl", 0.0), ("tree", "hybrid", 0.25), ("rmq", "hybrid", 0.5)])
def test_general_unknown_and_orphans_vs_oracle(api, method, strategy, factor):
    rng = np.random.default_rng(5)
    ids = np.array([1, 2, 3, 4, 5, 6, 7, 8, 20, 21, 22, 30], np.uint32)
    par = np.array([1, 1, 1, 2, 2, 3, 3, 4, 21, 99, 400, 1], np.uint32)
    tx = api.Taxonomy(ids, par)
    agg = O.Aggregator(O.Taxonomy.from_arrays(ids, par), method, strategy, factor)
    pool = np.array([1, 2, 3, 4, 5, 6, 7, 8, 20, 21, 22, 30, 9, 10, 50, 99, 100, 400, 5000], np.uint32)
    reads = _random_reads(rng, pool, 3000, 6, zero_frac=0.1)
    taxa, offs = _csr(reads)
    t, s = tx.aggregate(taxa, offs, method=method, strategy=strategy, factor=factor)
    nbad, bad, wt, ws = agg.taxa2agg_check(taxa, offs, t, s)
    assert nbad == 0, [(reads[int(b)], int(t[int(b)]), int(s[int(b)]), int(wt[int(b)]), int(ws[int(b)])) for b in bad[:5]]
    # the same reads at a fixed stride (the group kernel of the fused pipelines): a read with an unknown taxon next to resolved ones
    # must end in its error without its unresolved elements reaching the taxonomy tables (a GPU fault before the guard in P3b)
    import torch
    stride = 8
    strided = np.zeros((len(reads), stride), np.uint32)
    for i, r in enumerate(reads):
        strided[i, :len(r)] = r
    d_t = torch.empty(len(reads), dtype=torch.int32, device="cuda")
    d_s = torch.empty_like(d_t)
    tx.aggregate_strided_dev(torch.from_numpy(strided.view(np.int32)).cuda(), stride, len(reads), d_t, d_s, method=method, strategy=strategy, factor=factor)
    torch.cuda.synchronize()
    nbad, bad, _, _ = agg.taxa2agg_check(taxa, offs, d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32))
    assert nbad == 0, bad


def test_taxonomy_with_repeated_rows_vs_oracle(api):
    """A taxon id that occurs in several rows with the same parent: the reference keeps the last row's rank / valid (TaxonList::new,
    src/taxon.rs:135-145) and both child links (TaxonTree::new, :228-241); the device tables keep the first link with the last
    attributes.  Checked against the oracle, which restates the reference's structures literally (double links included), for every
    strategy and both snappings.  The same id under two different parents is refused."""
    from umgap_amd.synth import synthetic_taxonomy
    rng = np.random.default_rng(41)
    ids, par, valid, rank = synthetic_taxonomy(1500, 2500, max_depth=14, seed=9)
    ids2, par2, valid2, rank2 = list(ids), list(par), list(valid), list(rank)
    for p in rng.choice(len(ids), 25, replace=False):
        pos = int(rng.integers(p + 1, len(ids2) + 1))
        ids2.insert(pos, ids[p]); par2.insert(pos, par[p]); valid2.insert(pos, int(rng.integers(0, 2))); rank2.insert(pos, int(rng.integers(0, 28)))
    ids2, par2 = np.array(ids2, np.uint32), np.array(par2, np.uint32)
    valid2, rank2 = np.array(valid2, np.uint8), np.array(rank2, np.uint8)
    tx = api.Taxonomy(ids2, par2, valid2, rank2)
    otx = O.Taxonomy.from_arrays(ids2, par2, valid2, rank2)
    reads = _random_reads(rng, ids, 4000, 25, zero_frac=0.15)
    taxa, offs = _csr(reads)
    for method, strategy in (("tree", "lca*"), ("rmq", "lca*"), ("tree", "hybrid"), ("rmq", "hybrid"), ("rmq", "mrtl")):
        for ranked in (False, True):
            t, s = tx.aggregate(taxa, offs, method=method, strategy=strategy, factor=0.3, lower_bound=1.0, ranked_only=ranked)
            nbad, bad, _, _ = O.Aggregator(otx, method, strategy, 0.3).taxa2agg_check(taxa, offs, t, s, lower_bound=1.0, ranked=ranked)
            assert nbad == 0, (method, strategy, ranked, bad)
    with pytest.raises(api.UmgapError, match="different parents"):
        api.Taxonomy(np.array([1, 2, 3, 2], np.uint32), np.array([1, 1, 1, 3], np.uint32))


def test_mrtl_scored_vs_oracle(api):
    """rmq::rtl adds ancestor scores nearest-first (src/rmq/rtl.rs:42-46): deterministic up to ties."""
    from umgap_amd.synth import synthetic_taxonomy
    rng = np.random.default_rng(77)
    ids, par, valid, rank = synthetic_taxonomy(2000, 3000, max_depth=20, seed=5)
    tx = api.Taxonomy(ids, par, valid, rank)
    agg = O.Aggregator(O.Taxonomy.from_arrays(ids, par, valid, rank), "rmq", "mrtl")
    reads = []
    for _ in range(40):
        pool = rng.choice(ids, 6)
        reads += _random_reads(rng, pool, 40, 30)
    taxa, offs = _csr(reads)
    scores = (rng.integers(1, 64, len(taxa)) / 64.0).astype(np.float32)  # dyadic: sums exact in any order
    t, s = tx.aggregate(taxa, offs, method="rmq", strategy="mrtl", lower_bound=0.5, scores=scores)
    nbad, bad, _, _ = agg.taxa2agg_check(taxa, offs, t, s, scores=scores, lower_bound=0.5)
    assert nbad == 0, bad


# ---------------------------------------------------------------------------------------------
# seedextend on device vs the oracle's literal state machine
# ---------------------------------------------------------------------------------------------
def test_seedextend_vs_oracle(api, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(8)
    recs = []
    for _ in range(4000):
        n = int(rng.integers(0, 45))
        r = rng.choice(np.array([0, 0, 0, 5, 5, 5, 7, 9], np.uint32), n)
        recs.append(r)
    taxa, offs = _csr([r.tolist() for r in recs])
    for (s, g) in ((2, 0), (3, 1), (2, 1), (4, 1), (3, 2)):
        d_taxa = torch.from_numpy(taxa.view(np.int32).copy()).cuda()
        d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
        d_kept = torch.zeros(len(recs), dtype=torch.int32, device="cuda")
        api.seedextend_dev(d_taxa, d_offsets=d_offs, min_seed_size=s, max_gap_size=g, d_kept=d_kept)
        torch.cuda.synchronize()
        got = d_taxa.cpu().numpy().view(np.uint32)
        kept = d_kept.cpu().numpy()
        for i, r in enumerate(recs):
            want = O.seedextend_record(r, s, g)
            mine = got[int(offs[i]):int(offs[i + 1])]
            assert kept[i] == len(want), (i, r, want, mine)
            assert mine[mine != 0].tolist() == want[want != 0].tolist(), (i, r.tolist(), want, mine)


def test_seedextend_panic_cases_vs_oracle(api, torch_cuda):
    """--min-seed-size 0/1: a leading run of misses makes the reference slice a seed with start > end
    (src/commands/seedextend.rs:130-134 then :143-148 / :118, sliced at :169-173) and panic; the device kernel reports
    exactly those records with kept == NONE and prints nothing of them."""
    torch = torch_cuda
    rng = np.random.default_rng(18)
    recs = [np.array([0, 5], np.uint32), np.array([0, 0, 7], np.uint32), np.array([0], np.uint32), np.array([], np.uint32)]
    for _ in range(3000):
        n = int(rng.integers(0, 12))
        recs.append(rng.choice(np.array([0, 0, 0, 5, 5, 7], np.uint32), n))
    taxa, offs = _csr([r.tolist() for r in recs])
    n_panics = 0
    for (s, g) in ((1, 1), (0, 1), (1, 0), (1, 2), (0, 0), (2, 1)):
        d_taxa = torch.from_numpy(taxa.view(np.int32).copy()).cuda()
        d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
        d_kept = torch.zeros(len(recs), dtype=torch.int32, device="cuda")
        api.seedextend_dev(d_taxa, d_offsets=d_offs, min_seed_size=s, max_gap_size=g, d_kept=d_kept)
        torch.cuda.synchronize()
        got = d_taxa.cpu().numpy().view(np.uint32)
        kept = d_kept.cpu().numpy().view(np.uint32)
        for i, r in enumerate(recs):
            want = O.seedextend_record(r, s, g)
            mine = got[int(offs[i]):int(offs[i + 1])]
            if want is None:
                n_panics += 1
                assert kept[i] == 0xFFFFFFFF and not mine.any(), (s, g, r.tolist(), kept[i], mine)
            else:
                assert kept[i] == len(want), (s, g, r.tolist(), want, mine)
                assert mine[mine != 0].tolist() == want[want != 0].tolist(), (s, g, r.tolist(), want, mine)
    assert n_panics > 50     # the case is exercised; [0, 5] with -g1 -s1 is the hand-derived one
    assert O.seedextend_record(np.array([0, 5], np.uint32), 1, 1) is None


def test_seedextend_ranked_dev_vs_oracle(api, torch_cuda, tmp_path):
    """`seedextend -r taxa.tsv [-p penalty]` on the device (umgap_seedextend_ranked_dev; src/commands/seedextend.rs:77-83,150-163): of
    a record's extended seeds only the best-scoring one stays (the last of equal ones), the score from the ranks of the taxonomy
    the handle was made of.  Against the oracle's command on the same records as text: what it prints per record is what the
    device leaves, panics (-s 0 / 1) included; taxa the list does not hold and the unknown taxon 0 score the penalty."""
    torch = torch_cuda
    from umgap_amd.synth import synthetic_taxonomy, write_taxonomy_tsv
    ids, par, valid, rank = synthetic_taxonomy(300, 500, max_depth=8, seed=4)
    tsv = str(tmp_path / "taxons.tsv")
    write_taxonomy_tsv(tsv, ids, par, valid, rank)
    tx = api.Taxonomy(ids, par, valid, rank)
    rng = np.random.default_rng(28)
    pool = np.concatenate([ids[:12], [0, 0, 0, 0, 499, 100000]]).astype(np.uint32)      # 499 / 100000: not in the list
    recs = []
    for _ in range(3000):
        n = int(rng.integers(0, 40))
        r = rng.choice(pool, n)
        for _k in range(int(rng.integers(0, 4))):                                        # runs of equal taxa: seeds
            if n > 6:
                a = int(rng.integers(0, n - 3))
                r[a:a + int(rng.integers(2, 6))] = rng.choice(pool)
        recs.append(r.astype(np.uint32))
    recs += [np.array([], np.uint32), np.array([0], np.uint32), np.array([0, 5], np.uint32)]   # (the last one panics with -s 1 -g 1)
    # inverted seeds (start > end, src/commands/seedextend.rs:119 under -s 1): `.take(e - s)` (:158) wraps in a release build and the
    # seed scores every taxon from s on -- the record VERDICT r4 names, and short records rich in zero-runs
    five = int(ids[5])
    recs += [np.array([0, 7, 0, 0, 0, five, five, five], np.uint32), np.array([0, five, 0, 0, five, five], np.uint32)]
    small = np.array([five, int(ids[7]), 0, 0, 0, 499], np.uint32)
    recs += [rng.choice(small, int(rng.integers(1, 14))).astype(np.uint32) for _ in range(1500)]
    taxa, offs = _csr([r.tolist() for r in recs])
    text = b"".join(b">r%d\n" % i + b"".join(b"%d\n" % t for t in r) for i, r in enumerate(recs))
    ranked = O.RankedList(O.Taxonomy.from_arrays(ids, par, valid, rank))
    n_panics = n_kept = n_release_only = 0
    for (s, g, pen) in ((2, 0, 5), (3, 1, 5), (2, 2, 0), (1, 1, 5), (1, 1, 0), (1, 2, 1), (0, 1, 30), (4, 1, 100)):
        d_taxa = torch.from_numpy(taxa.view(np.int32).copy()).cuda()
        d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
        d_kept = torch.zeros(len(recs), dtype=torch.int32, device="cuda")
        api.seedextend_ranked_dev(tx, d_taxa, d_offs, min_seed_size=s, max_gap_size=g, penalty=pen, d_kept=d_kept)
        torch.cuda.synchronize()
        got = d_taxa.cpu().numpy().view(np.uint32)
        kept = d_kept.cpu().numpy().view(np.uint32)
        # the oracle record by record (a release build; the command itself dies at its first panicking record, below)
        for i, r in enumerate(recs):
            mine = got[int(offs[i]):int(offs[i + 1])]
            want = ranked.seedextend_record(r, s, g, pen)
            if want is None:
                n_panics += 1
                assert kept[i] == 0xFFFFFFFF and not mine.any(), (s, g, pen, r.tolist(), kept[i])
                continue
            if ranked.seedextend_record(r, s, g, pen, overflow_checks=True) is None:
                n_release_only += 1                                  # an inverted seed that lost: only a debug build panics here
            assert kept[i] == len(want), (s, g, pen, i, r.tolist(), want.tolist(), mine.tolist())
            assert mine[mine != 0].tolist() == [t for t in want.tolist() if t], (s, g, pen, i, r.tolist(), want.tolist(), mine.tolist())
            n_kept += len(want) > 0
        # ... and the oracle's command on the same records as text agrees with its record-by-record form up to its first panic
        orc, out, _ = O.run_cli(["seedextend", "-s", str(s), "-g", str(g), "-r", tsv, "-p", str(pen)], text)
        assert orc in (0, 101), (s, g, orc)
        n_out = out.count(b">")
        first_panic = next((i for i, r in enumerate(recs) if ranked.seedextend_record(r, s, g, pen) is None), None)
        assert (orc == 101) == (first_panic is not None) and n_out == (len(recs) if first_panic is None else first_panic), (s, g, pen, orc, n_out, first_panic)
    assert n_kept > 3000 and n_panics >= 10 and n_release_only >= 1


# ---------------------------------------------------------------------------------------------
# the synthetic 9-mer high-precision pipeline: lookup -> (seedextend) -> LCA* with -l 2
# ---------------------------------------------------------------------------------------------
def _pipeline_check(api, torch, n_tax, n_keys, n_pairs, with_seedextend, w=None):
    from umgap_amd.synth import ADD_B, KMERS_PER_PAIR, MUL_A, SyntheticWorkload
    w = w or SyntheticWorkload(n_tax=n_tax, n_keys=n_keys)
    d_keys = w.reads(0, n_pairs)
    d_hits = w.index.lookup9_dev(d_keys, miss_value=0)
    torch.cuda.synchronize()
    keys = d_keys.cpu().numpy().view(np.uint64)
    hits = d_hits.cpu().numpy().view(np.uint32)
    # (1) lookup parity against the closed-form oracle index
    p = O.SynthIndexParams(n_keys, n_tax, MUL_A, ADD_B)
    want_hits = O.synth_lookup9(p, w.taxon_of_row, keys, miss_value=0)
    assert (hits == want_hits).all()
    frac_hit = (hits != 0).mean()
    assert 0.5 < frac_hit < 0.8, frac_hit
    # (2) seedextend per fragment (42 k-mers), in place
    if with_seedextend:
        half = KMERS_PER_PAIR // 2
        api.seedextend_dev(d_hits, stride=half, n_records=2 * n_pairs, min_seed_size=3, max_gap_size=1)
        torch.cuda.synchronize()
        se = d_hits.cpu().numpy().view(np.uint32)
        for r in range(0, 2 * n_pairs, max(1, (2 * n_pairs) // 500)):
            want = O.seedextend_record(want_hits[r * half:(r + 1) * half], 3, 1)
            mine = se[r * half:(r + 1) * half]
            assert mine[mine != 0].tolist() == want[want != 0].tolist()
        want_hits = se
    # (3) aggregate per pair: taxa2agg -l2 -a lca*  (scripts/umgap-analyse.sh:304)
    d_t = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    d_s = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    w.taxonomy.aggregate_strided_dev(d_hits, KMERS_PER_PAIR, n_pairs, d_t, d_s, method="tree", strategy="lca*", lower_bound=2.0)
    torch.cuda.synchronize()
    ot = O.Taxonomy.from_arrays(w.ids, w.parents, w.valid, w.rank)
    agg = O.Aggregator(ot, "tree", "lca*")
    offs = np.arange(n_pairs + 1, dtype=np.uint64) * KMERS_PER_PAIR
    t = d_t.cpu().numpy().view(np.uint32)
    s = d_s.cpu().numpy().view(np.uint32)
    nbad, bad, wt, ws = agg.taxa2agg_check(want_hits, offs, t, s, lower_bound=2.0, nthreads=8)
    assert nbad == 0, (bad, t[bad.astype(int)], wt[bad.astype(int)])
    return t, s


@pytest.mark.parametrize("with_seedextend", [False, True])
@pytest.mark.parametrize("stride,n_pairs", [(84, 20_001), (42, 5_000), (128, 3_001), (7, 1_000)])
def test_fused_classify_vs_oracle(api, torch_cuda, stride, n_pairs, with_seedextend):
    """umgap_classify9_dev (probe -> [seedextend] -> LCA* in one kernel) against the oracle stages."""
    torch = torch_cuda
    from umgap_amd.synth import ADD_B, MUL_A, SyntheticWorkload
    if with_seedextend and stride % 2:
        pytest.skip("two fragments need an even stride")
    n_tax, n_keys = 20_000, 1_000_000
    w = SyntheticWorkload(n_tax=n_tax, n_keys=n_keys)
    d_keys = w.reads(0, n_pairs, kmers_per_pair=stride)
    d_t = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    d_s = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    api.classify9_dev(w.index, w.taxonomy, d_keys, n_pairs, stride, d_t, d_s, method="tree", lower_bound=2.0,
                      fragments_per_read=2 if with_seedextend else 0, min_seed_size=3, max_gap_size=1)
    torch.cuda.synchronize()
    keys = d_keys.cpu().numpy().view(np.uint64)
    hits = O.synth_lookup9(O.SynthIndexParams(n_keys, n_tax, MUL_A, ADD_B), w.taxon_of_row, keys, 0)
    if with_seedextend:
        half = stride // 2
        out = np.zeros_like(hits)
        for r in range(2 * n_pairs):
            kept = O.seedextend_record(hits[r * half:(r + 1) * half], 3, 1)
            out[r * half:r * half + len(kept)] = kept      # order within the record is irrelevant to taxa2agg
        hits = out
    agg = O.Aggregator(O.Taxonomy.from_arrays(w.ids, w.parents, w.valid, w.rank), "tree", "lca*")
    offs = np.arange(n_pairs + 1, dtype=np.uint64) * stride
    t, s = d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32)
    nbad, bad, wt, ws = agg.taxa2agg_check(hits, offs, t, s, lower_bound=2.0, nthreads=8)
    assert nbad == 0, (bad, t[bad.astype(int)], wt[bad.astype(int)])
    # and it agrees with the unfused device path bit for bit
    d_hits = w.index.lookup9_dev(d_keys)
    if with_seedextend:
        api.seedextend_dev(d_hits, stride=stride // 2, n_records=2 * n_pairs, min_seed_size=3, max_gap_size=1)
    d_t2, d_s2 = torch.empty_like(d_t), torch.empty_like(d_s)
    w.taxonomy.aggregate_strided_dev(d_hits, stride, n_pairs, d_t2, d_s2, method="tree", strategy="lca*", lower_bound=2.0)
    torch.cuda.synchronize()
    assert torch.equal(d_t, d_t2) and torch.equal(d_s, d_s2)


@pytest.mark.parametrize("method,strategy", [("tree", "lca*"), ("tree", "hybrid"), ("rmq", "mrtl")])
def test_fused_classify_seedextend_panic_status(api, torch_cuda, method, strategy):
    """--min-seed-size 1: where the reference's seedextend would panic on a fragment (src/commands/seedextend.rs:169-173,
    exit 101) the fused kernels report UMGAP_READ_PANIC for the read; every other read is aggregated as usual."""
    torch = torch_cuda
    from umgap_amd.synth import ADD_B, MUL_A, SyntheticWorkload
    n_tax, n_keys, stride, n_pairs = 20_000, 1_000_000, 12, 6_000
    w = SyntheticWorkload(n_tax=n_tax, n_keys=n_keys)
    d_keys = w.reads(0, n_pairs, kmers_per_pair=stride)
    d_t = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    d_s = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    api.classify9_dev(w.index, w.taxonomy, d_keys, n_pairs, stride, d_t, d_s, method=method, strategy=strategy, factor=0.25, lower_bound=1.0,
                      fragments_per_read=2, min_seed_size=1, max_gap_size=1)
    torch.cuda.synchronize()
    keys = d_keys.cpu().numpy().view(np.uint64)
    hits = O.synth_lookup9(O.SynthIndexParams(n_keys, n_tax, MUL_A, ADD_B), w.taxon_of_row, keys, 0)
    half = stride // 2
    t, s = d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32)
    panics, ok_reads, ok_rows = 0, [], []
    for r in range(n_pairs):
        frs = [O.seedextend_record(hits[(2 * r + f) * half:(2 * r + f + 1) * half], 1, 1) for f in (0, 1)]
        if any(f is None for f in frs):
            panics += 1
            assert s[r] == 3 and t[r] == 0, (r, s[r], t[r])
        else:
            ok_reads.append(r)
            ok_rows.append(np.concatenate(frs).tolist())
    assert panics > 20 and len(ok_reads) > 1000, (panics, len(ok_reads))
    taxa, offs = _csr(ok_rows)
    agg = O.Aggregator(O.Taxonomy.from_arrays(w.ids, w.parents, w.valid, w.rank), method, strategy, 0.25)
    idx = np.array(ok_reads)
    nbad, bad, _, _ = agg.taxa2agg_check(taxa, offs, t[idx].copy(), s[idx].copy(), lower_bound=1.0, nthreads=8)
    assert nbad == 0, bad


@pytest.mark.parametrize("with_seedextend", [False, True])
@pytest.mark.parametrize("method,strategy,factor,lb", [("tree", "hybrid", 0.25, 1.0), ("rmq", "mrtl", 0.25, 1.0), ("rmq", "hybrid", 0.5, 0.0),
                                                        ("tree", "hybrid", 0.8, 2.0), ("rmq", "mrtl", 0.25, 5.0)])
@pytest.mark.parametrize("stride,n_pairs", [(84, 20_001), (128, 2_001), (6, 999)])
def test_fused_classify_other_strategies(api, torch_cuda, stride, n_pairs, method, strategy, factor, lb, with_seedextend):
    """umgap_classify9_strategy_dev (probe -> [seedextend] -> MRTL / hybrid in one kernel, the sensitivity pipelines of
    scripts/umgap-analyse.sh:277-288) against the oracle (tie sets) and, bit for bit, against the unfused device path."""
    torch = torch_cuda
    from umgap_amd.synth import ADD_B, MUL_A, SyntheticWorkload
    n_tax, n_keys = 20_000, 1_000_000
    w = SyntheticWorkload(n_tax=n_tax, n_keys=n_keys)
    d_keys = w.reads(0, n_pairs, kmers_per_pair=stride)
    d_t = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    d_s = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    api.classify9_dev(w.index, w.taxonomy, d_keys, n_pairs, stride, d_t, d_s, method=method, strategy=strategy, factor=factor, lower_bound=lb,
                      fragments_per_read=2 if with_seedextend else 0, min_seed_size=2, max_gap_size=1)
    d_hits = w.index.lookup9_dev(d_keys)
    if with_seedextend:
        api.seedextend_dev(d_hits, stride=stride // 2, n_records=2 * n_pairs, min_seed_size=2, max_gap_size=1)
    d_t2, d_s2 = torch.empty_like(d_t), torch.empty_like(d_s)
    w.taxonomy.aggregate_strided_dev(d_hits, stride, n_pairs, d_t2, d_s2, method=method, strategy=strategy, factor=factor, lower_bound=lb)
    torch.cuda.synchronize()
    assert torch.equal(d_t, d_t2) and torch.equal(d_s, d_s2)
    agg = O.Aggregator(O.Taxonomy.from_arrays(w.ids, w.parents, w.valid, w.rank), method, strategy, factor=factor)
    hits = d_hits.cpu().numpy().view(np.uint32)
    offs = np.arange(n_pairs + 1, dtype=np.uint64) * stride
    t, s = d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32)
    nbad, bad, wt, ws = agg.taxa2agg_check(hits, offs, t, s, lower_bound=lb, nthreads=8)
    assert nbad == 0, (bad, t[bad.astype(int)], wt[bad.astype(int)])
    assert (s == 0).any()
    # and on the table bound to the taxonomy (interval codes instead of ids): bit-identical
    w.index.bind(w.taxonomy)
    d_t3, d_s3 = torch.empty_like(d_t), torch.empty_like(d_s)
    api.classify9_dev(w.index, w.taxonomy, d_keys, n_pairs, stride, d_t3, d_s3, method=method, strategy=strategy, factor=factor, lower_bound=lb,
                      fragments_per_read=2 if with_seedextend else 0, min_seed_size=2, max_gap_size=1)
    torch.cuda.synchronize()
    w.index.unbind(w.taxonomy)
    assert torch.equal(d_t, d_t3) and torch.equal(d_s, d_s3)


def test_fused_classify_other_strategies_many_distinct_taxa(api, torch_cuda):
    """Reads with more than 64 distinct taxa among their 128 k-mers take the second launch (larger scratch)."""
    torch = torch_cuda
    from umgap_amd.api import Index, Taxonomy
    from umgap_amd.synth import synthetic_taxonomy
    ids, par, valid, rank = synthetic_taxonomy(5000, 8000, max_depth=12, seed=9)
    tx = Taxonomy(ids, par, valid, rank)
    rng = np.random.default_rng(3)
    n_keys, stride, n_reads = 40_000, 128, 600
    keys = rng.permutation(np.unique(rng.integers(0, 1 << 45, 2 * n_keys, dtype=np.uint64)))[:n_keys]
    vals = rng.choice(ids, n_keys).astype(np.uint32)
    ix = Index.build(keys, vals)
    q = keys[rng.integers(0, n_keys, n_reads * stride)]            # all hits, ~128 distinct taxa per read
    q[: 100 * stride] = keys[rng.integers(0, 40, 100 * stride)]   # the first 100 reads: few distinct taxa
    d_keys = torch.from_numpy(q.view(np.int64)).cuda()
    d_t = torch.empty(n_reads, dtype=torch.int32, device="cuda")
    d_s = torch.empty_like(d_t)
    d_t2, d_s2 = torch.empty_like(d_t), torch.empty_like(d_t)
    for method, strategy in (("tree", "hybrid"), ("rmq", "mrtl"), ("rmq", "hybrid")):
        api.classify9_dev(ix, tx, d_keys, n_reads, stride, d_t, d_s, method=method, strategy=strategy, factor=0.3, lower_bound=0.0)
        d_hits = ix.lookup9_dev(d_keys)
        tx.aggregate_strided_dev(d_hits, stride, n_reads, d_t2, d_s2, method=method, strategy=strategy, factor=0.3, lower_bound=0.0)
        torch.cuda.synchronize()
        assert torch.equal(d_t, d_t2) and torch.equal(d_s, d_s2)
        hits = d_hits.cpu().numpy().view(np.uint32)
        assert max(len(set(hits[r * stride:(r + 1) * stride].tolist())) for r in range(100, n_reads)) > 64
        agg = O.Aggregator(O.Taxonomy.from_arrays(ids, par, valid, rank), method, strategy, factor=0.3)
        offs = np.arange(n_reads + 1, dtype=np.uint64) * stride
        nbad, bad, _, _ = agg.taxa2agg_check(hits, offs, d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32), lower_bound=0.0, nthreads=8)
        assert nbad == 0, bad


@pytest.mark.parametrize("stride,n_pairs,lb", [(84, 30_001, 2.0), (84, 10_000, 1.0), (128, 3_001, 0.0), (42, 5_000, 3.0), (7, 1_000, 1.0)])
def test_bound_table_classify_is_bit_identical(api, torch_cuda, stride, n_pairs, lb):
    """umgap_index_bind_taxonomy rewrites the table values into {preorder, subtree-size code}; the fused kernel then skips
    the by-id gathers.  Same answers, bit for bit, as the unbound kernel (which is checked against the oracle above), with
    and without seedextend, for both methods and -r; plain lookups are refused while bound; unbinding restores the ids."""
    torch = torch_cuda
    from umgap_amd import UmgapError
    from umgap_amd.synth import SyntheticWorkload
    w = SyntheticWorkload(n_tax=60_000, n_keys=2_000_000, max_depth=30)
    d_keys = w.reads(0, n_pairs, kmers_per_pair=stride)
    want_hits = w.index.lookup9_dev(d_keys).clone()
    outs = {}
    cases = [(m, r, f) for m in ("tree", "rmq") for r in (False, True) for f in ((0, 2) if stride % 2 == 0 else (0,))]
    for m, r, f in cases:
        d_t = torch.empty(n_pairs, dtype=torch.int32, device="cuda"); d_s = torch.empty_like(d_t)
        api.classify9_dev(w.index, w.taxonomy, d_keys, n_pairs, stride, d_t, d_s, method=m, lower_bound=lb, ranked_only=r,
                          fragments_per_read=f, min_seed_size=2, max_gap_size=1)
        outs[(m, r, f)] = (d_t, d_s)
    w.index.bind(w.taxonomy)
    assert w.index.bound
    try:
        for m, r, f in cases:
            d_t = torch.empty(n_pairs, dtype=torch.int32, device="cuda"); d_s = torch.empty_like(d_t)
            api.classify9_dev(w.index, w.taxonomy, d_keys, n_pairs, stride, d_t, d_s, method=m, lower_bound=lb, ranked_only=r,
                              fragments_per_read=f, min_seed_size=2, max_gap_size=1)
            torch.cuda.synchronize()
            assert torch.equal(d_t, outs[(m, r, f)][0]) and torch.equal(d_s, outs[(m, r, f)][1]), (m, r, f)
        with pytest.raises(UmgapError):
            w.index.lookup9_dev(d_keys)
        with pytest.raises(UmgapError):
            w.index.bind(w.taxonomy)
    finally:
        w.index.unbind(w.taxonomy)
    assert not w.index.bound
    assert torch.equal(w.index.lookup9_dev(d_keys), want_hits)
    assert (outs[("tree", False, 0)][1] == 0).any()


@pytest.mark.parametrize("shape", ["caterpillar", "broom", "random", "binary"])
def test_bound_table_adversarial_tree_shapes(api, torch_cuda, shape):
    """The 10-bit subtree-size code is exact below 32 and rounded above: trees whose subtree sizes sit around the
    rounding boundaries (long chains with side leaves, brooms, complete binary trees) hit the 'exact last' path often.
    Bound vs unbound vs oracle, reads = small random sets of taxa."""
    torch = torch_cuda
    rng = np.random.default_rng({"caterpillar": 1, "broom": 2, "random": 3, "binary": 4}[shape])
    n = 6000
    ids = np.arange(1, n + 1, dtype=np.uint32)
    par = np.ones(n, np.uint32)
    if shape == "caterpillar":      # a spine with a leaf hanging off every spine node
        for i in range(1, n):
            par[i] = (i - 1 if i % 2 == 0 else i) if i > 1 else 1
            par[i] = max(1, (i // 2) * 2 - 1) if i % 2 == 0 else max(1, i - 2)
    elif shape == "broom":          # a long handle, then a wide fan
        for i in range(1, n):
            par[i] = i if i < 3000 else 3000
    elif shape == "binary":
        for i in range(1, n):
            par[i] = (i + 1) // 2
    else:
        for i in range(1, n):
            par[i] = int(rng.integers(max(1, i - 50), i + 1))
    par[0] = 1
    tx = api.Taxonomy(ids, par)
    keys = np.arange(1000, 1000 + n, dtype=np.uint64) * np.uint64(7919)
    ix = api.Index.build(keys, ids)
    n_reads, stride = 4000, 8
    pick = rng.integers(0, n, (n_reads, stride))
    pick[:, 4:] = pick[:, :4]                       # every taxon twice, so -l 2 keeps them
    q = keys[pick.reshape(-1)]
    d_keys = torch.from_numpy(q.view(np.int64)).cuda()
    agg = O.Aggregator(O.Taxonomy.from_arrays(ids, par), "tree", "lca*")
    offs = np.arange(n_reads + 1, dtype=np.uint64) * stride
    hits = ids[pick.reshape(-1)]
    res = {}
    for bound in (False, True):
        if bound:
            ix.bind(tx)
        d_t = torch.empty(n_reads, dtype=torch.int32, device="cuda"); d_s = torch.empty_like(d_t)
        api.classify9_dev(ix, tx, d_keys, n_reads, stride, d_t, d_s, method="tree", lower_bound=2.0)
        torch.cuda.synchronize()
        res[bound] = (d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32))
    ix.unbind(tx)
    assert (res[False][0] == res[True][0]).all() and (res[False][1] == res[True][1]).all()
    nbad, bad, _, _ = agg.taxa2agg_check(hits, offs, res[True][0], res[True][1], lower_bound=2.0, nthreads=8)
    assert nbad == 0, bad


def test_bind_refuses_values_outside_the_taxonomy(api, torch_cuda):
    """A table value that is not a reachable taxon has no interval: binding is refused and the table is left untouched."""
    from umgap_amd import UmgapError
    ids = np.arange(1, 40, dtype=np.uint32)
    par = np.maximum(1, ids // 2).astype(np.uint32)
    par[0] = 1
    tx = api.Taxonomy(ids, par)
    rng = np.random.default_rng(1)
    keys = np.unique(rng.integers(0, 1 << 45, 5000, dtype=np.uint64))
    vals = rng.choice(ids, len(keys)).astype(np.uint32)
    vals[17] = 7_000_001
    ix = api.Index.build(keys, vals)
    with pytest.raises(UmgapError):
        ix.bind(tx)
    assert not ix.bound
    import torch
    got = ix.lookup9_dev(torch.from_numpy(keys.view(np.int64)).cuda()).cpu().numpy().view(np.uint32)
    assert (got == vals).all()
    vals[17] = 3
    ix2 = api.Index.build(keys, vals)
    ix2.bind(tx)
    ix2.unbind(tx)
    got = ix2.lookup9_dev(torch.from_numpy(keys.view(np.int64)).cuda()).cpu().numpy().view(np.uint32)
    assert (got == vals).all()


def test_synth_pipeline_small(api, torch_cuda):
    t, s = _pipeline_check(api, torch_cuda, n_tax=20_000, n_keys=1_000_000, n_pairs=20_000, with_seedextend=False)
    assert (s == 0).mean() > 0.9 and len(np.unique(t)) > 1000


def test_synth_pipeline_small_with_seedextend(api, torch_cuda):
    _pipeline_check(api, torch_cuda, n_tax=20_000, n_keys=1_000_000, n_pairs=20_000, with_seedextend=True)


# (BASELINE.json configs[1], 1 M pairs against the FULL 8 G-key index, both surfaces: with the full-size tests at the end of the file)


# ---------------------------------------------------------------------------------------------
# sharded mode: shard-local builds + partition kernel, with the all-to-all emulated in one process
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_sharded_components_single_gpu(api, torch_cuda, n_shards):
    torch = torch_cuda
    from umgap_amd import _lib
    from umgap_amd.api import _dev_ptr, _stream_ptr
    from umgap_amd.synth import ADD_B, MUL_A, SyntheticWorkload
    n_tax, n_keys, n_pairs = 5000, 600_000, 3000
    w = SyntheticWorkload(n_tax=n_tax, n_keys=n_keys)                      # whole table: the truth
    keys = w.reads(0, n_pairs)
    want = w.index.lookup9_dev(keys).clone()
    shards = [api.Index.synthetic(n_keys, w.d_taxon_of_row, MUL_A, ADD_B, 0.0, s, n_shards) for s in range(n_shards)]
    assert abs(sum(sh.table_bytes for sh in shards) - w.index.table_bytes) < n_shards * 6 * 64 * 2
    K = keys.numel()
    cap = int(K / n_shards * 1.2) + 4096
    send = torch.full((n_shards * cap,), -1, dtype=torch.int64, device="cuda")
    perm = torch.empty(K, dtype=torch.int32, device="cuda")
    cursor = torch.zeros(n_shards, dtype=torch.int64, device="cuda")
    overflow = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.load().umgap_index_partition9_dev(shards[0]._h, _dev_ptr(keys), K, n_shards, cap, _dev_ptr(send), _dev_ptr(perm),
                                                      _dev_ptr(cursor), _dev_ptr(overflow), _stream_ptr(None)))
    torch.cuda.synchronize()
    assert int(overflow.item()) == 0 and int(cursor.sum().item()) == K
    assert cursor.min().item() > 0.8 * K / n_shards        # the mixer spreads the keys evenly
    owner = w.index.owner9_dev(keys, n_shards)
    torch.cuda.synchronize()
    assert (owner.cpu().numpy() == (perm // cap).cpu().numpy()).all()
    # "all-to-all": region s is what shard s would receive; its answers travel back in place
    back = torch.empty(n_shards * cap, dtype=torch.int32, device="cuda")
    for s in range(n_shards):
        shards[s].lookup9_dev(send[s * cap:(s + 1) * cap], back[s * cap:(s + 1) * cap])
    hits = torch.index_select(back, 0, perm)
    torch.cuda.synchronize()
    assert (hits.cpu().numpy() == want.cpu().numpy()).all()
    # a key sent to the wrong shard is a miss there, never a wrong taxon
    wrong = shards[0].lookup9_dev(send[cap:2 * cap][: int(cursor[1].item())])
    torch.cuda.synchronize()
    assert (wrong.cpu().numpy() == 0).all()


# ---------------------------------------------------------------------------------------------
# variable-length (tryptic) index: exact byte-string keys
# ---------------------------------------------------------------------------------------------
def _random_peptides(rng, n, lo=5, hi=50):
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    out = set()
    while len(out) < n:
        k = int(rng.integers(lo, hi + 1))
        out.add(rng.choice(letters, k).tobytes())
    return sorted(out)


@pytest.mark.parametrize("layout", ["keys in the slots", "slots + heap"])
def test_var_index_matches_dict(api, tmp_path, layout, monkeypatch):
    monkeypatch.setenv("UMGAP_VAR_PACKED", "1" if layout == "keys in the slots" else "0")   # read at build time
    rng = np.random.default_rng(12)
    peps = _random_peptides(rng, 120_000)
    vals = rng.integers(1, 1 << 31, len(peps)).astype(np.uint32)
    d = dict(zip(peps, vals.tolist()))
    for load in (0.0, 0.9):       # 0.9 forces overflow chains
        ix = api.Index.build_var(peps, vals, load_factor=load)
        assert ix.kind == 2 and ix.num_keys == len(peps)
        assert ix.var_layout == (1 if layout == "keys in the slots" else 0)
        assert (ix.lookup_var(peps) == vals).all()
        q = peps[:3000] + _random_peptides(rng, 3000) + [b"", b"A" * 64, b"K", peps[0][:-1], peps[0] + b"A", b"ACD*F", b"\xff\x00\x01"]
        want = np.array([d.get(p, 0) for p in q], np.uint32)
        assert (ix.lookup_var(q, miss_value=0) == want).all()
        want9 = np.array([d.get(p, 9) for p in q], np.uint32)
        assert (ix.lookup_var(q, miss_value=9) == want9).all()
    path = tmp_path / "tryptic.umgapix"
    ix.save(path)
    ix2 = api.Index.open(path)
    assert ix2.kind == 2 and ix2.var_layout == ix.var_layout and (ix2.lookup_var(q) == want).all()
    assert len(ix.lookup_var([])) == 0
    with pytest.raises(api.UmgapError):
        ix.lookup9(np.zeros(3, np.uint64))
    with pytest.raises(api.UmgapError, match="duplicate"):
        api.Index.build_var([b"AAAAA", b"CCCCC", b"AAAAA"], np.array([1, 2, 3], np.uint32))
    with pytest.raises(api.UmgapError, match="63"):
        api.Index.build_var([b"A" * 64], np.array([1], np.uint32))


def test_var_index_inline_keys(api, tmp_path):
    """Keys of at most 17 letters live inside their slot (index_var.hip): lengths around the limit, residues that straddle the
    slot's words, 'A' padding told apart by the length, short keys that do NOT qualify (bytes outside 'A'..'Z'), full tables
    with overflow chains and the stash -- against a dict."""
    rng = np.random.default_rng(99)
    letters = np.frombuffer(b"ABCDEFGHIKLMNPQRSTVWYZ", np.uint8)
    special = [b"A", b"AA", b"AAA", b"A" * 16, b"A" * 17, b"A" * 18, b"Z", b"Z" * 17, b"Z" * 18, b"ACDEFGHIKLMNPQRST", b"ACDEFGHIKLMNPQRSTV",
               b"abc", b"A*C", b"\x00", b"\xff" * 5, b"[", b"@", b"AAAAAAAAAAAAZ", b"AAAAAAAAAAAAP", b"AAAAAAAAAAAAAZ", b"AAAAAAAAAAAABA"]
    keys = set(special)
    for n in range(1, 22):
        for _ in range(3000 if n > 4 else 20):
            keys.add(rng.choice(letters, n).tobytes())
    for _ in range(2000):                                        # short keys with one byte that does not qualify
        k = bytearray(rng.choice(letters, int(rng.integers(1, 18))).tobytes())
        k[int(rng.integers(0, len(k)))] = int(rng.choice([42, 64, 91, 97, 0, 255]))
        keys.add(bytes(k))
    keys = sorted(keys)
    vals = rng.integers(1, 1 << 31, len(keys)).astype(np.uint32)
    d = dict(zip(keys, vals.tolist()))
    strangers = [rng.choice(letters, int(rng.integers(1, 22))).tobytes() for _ in range(20000)]
    near = [k[:-1] for k in keys[:4000] if len(k) > 1] + [k + b"A" for k in keys[:4000]] + [k[:-1] + b"B" for k in keys[4000:8000]]
    q = keys + strangers + near
    want = np.array([d.get(p, 0) for p in q], np.uint32)
    for load in (0.0, 0.93):
        ix = api.Index.build_var(keys, vals, load_factor=load)
        assert (ix.lookup_var(q, miss_value=0) == want).all(), load
    path = tmp_path / "inline.umgapix"
    ix.save(path)
    assert (api.Index.open(path).lookup_var(q) == want).all()
    with pytest.raises(api.UmgapError, match="duplicate"):
        api.Index.build_var([b"AAAAAK", b"CCCCC", b"AAAAAK"], np.array([1, 2, 3], np.uint32))


def test_var_index_keys_in_the_slots(api, tmp_path):
    """The packed layout (index_varp.hip): a key takes 1, 2 or 3 adjacent slots of one bucket (16 residues in the header slot, 24
    per continuation).  Lengths around every boundary, keys that differ only in a continuation slot or only in their length
    ('A' is residue 0, so padding and a trailing A must be told apart by the length), continuation images that are all zero,
    queries with bytes outside 'A'..'Z' (never in such a table), full tables with overflow chains and the stash, save / open,
    and the dump (printindex) -- against a dict."""
    rng = np.random.default_rng(321)
    letters = np.frombuffer(b"ABCDEFGHIKLMNPQRSTVWYZ", np.uint8)
    keys = set()
    for n in list(range(1, 64)):
        for _ in range(400 if n > 3 else 15):
            keys.add(rng.choice(letters, n).tobytes())
    for n in (1, 15, 16, 17, 18, 39, 40, 41, 42, 62, 63):
        keys.add(b"A" * n)
        keys.add(b"Z" * n)
        keys.add(b"A" * (n - 1) + b"B")
    base = [rng.choice(letters, 63).tobytes() for _ in range(300)]
    for k in base:                                   # families that share every slot but one
        for cut in (16, 17, 40, 41, 63):
            keys.add(k[:cut])
        keys.add(k[:62] + (b"A" if k[62:] != b"A" else b"B"))
        keys.add(k[:30] + b"A" + k[31:])
        keys.add(k[:5] + b"A" + k[6:])
    keys = sorted(keys)
    vals = rng.integers(1, 1 << 32, len(keys), dtype=np.uint64).astype(np.uint32)
    d = dict(zip(keys, vals.tolist()))
    strangers = [rng.choice(letters, int(rng.integers(1, 64))).tobytes() for _ in range(20000)]
    near = [k[:-1] for k in keys[:6000] if len(k) > 1] + [k + b"A" for k in keys[:6000] if len(k) < 63] + [k[:-1] + b"C" for k in keys[6000:12000]]
    odd = [b"", b"A" * 64, b"A" * 200, b"abc", b"ACD*F", b"\xff\x00\x01", b"A" * 20 + b"*" + b"A" * 10, b"[" * 17, b"@" * 41, keys[5][:-1] + b"a"]
    q = keys + strangers + near + odd
    want = np.array([d.get(p, 0) for p in q], np.uint32)
    for load in (0.0, 0.3, 0.95):
        ix = api.Index.build_var(keys, vals, load_factor=load)
        assert ix.var_layout == 1
        assert (ix.lookup_var(q, miss_value=0) == want).all(), load
        assert (ix.lookup_var(q[::-1], miss_value=7) == np.array([d.get(p, 7) for p in q[::-1]], np.uint32)).all(), load
    path = tmp_path / "packed.umgapix"
    ix.save(path)
    ix2 = api.Index.open(path)
    assert ix2.var_layout == 1 and ix2.num_keys == len(keys) and (ix2.lookup_var(q) == want).all()
    # the dump gives every key back, once
    from umgap_amd import _lib
    import ctypes as C
    n, hb = C.c_size_t(), C.c_uint64()
    cap = len(keys) + 10
    offs, lens, taxa = np.zeros(cap, np.uint64), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    L = _lib.load()
    _lib.check(L.umgap_index_dump_var(ix2._h, None, 0, offs.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), taxa.ctypes.data_as(C.c_void_p), cap, C.byref(n), C.byref(hb)))
    heap = np.zeros(hb.value + 1, np.uint8)
    _lib.check(L.umgap_index_dump_var(ix2._h, heap.ctypes.data_as(C.c_void_p), hb.value, offs.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), taxa.ctypes.data_as(C.c_void_p), cap, C.byref(n), C.byref(hb)))
    assert n.value == len(keys)
    got = {heap[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes(): int(taxa[i]) for i in range(n.value)}
    assert got == d
    with pytest.raises(api.UmgapError, match="duplicate"):
        api.Index.build_var([b"AAAAAK" * 5, b"CCCCC", b"AAAAAK" * 5], np.array([1, 2, 3], np.uint32))


def test_pept2lca_doc_example_tryptic(api):
    """src/commands/pept2lca.rs:24-46 verbatim."""
    ix = api.Index.build_var([b"AAALTER", b"ENFVYLAK"], np.array([2, 3398], np.uint32))
    assert ix.lookup_var([b"AAALTER", b"ENFVYLAK"]).tolist() == [2, 3398]
    assert ix.lookup_var([b"NOTATRYPTICPEPTIDE", b"ENFVYLAK"], miss_value=0).tolist() == [0, 3398]


def test_fused_kernels_refuse_a_var_index(api, torch_cuda):
    """A variable-length (kind 2) handle carries no 9-mer table: every fused 9-mer entry point must answer
    UMGAP_ERR_INVALID instead of probing whatever its view holds."""
    torch = torch_cuda
    from umgap_amd.synth import SyntheticWorkload
    w = SyntheticWorkload(n_tax=2_000, n_keys=10_000)
    ix = api.Index.build_var([b"AAALTER", b"ENFVYLAK"], np.array([2, 3398], np.uint32))
    d_keys = torch.zeros(84 * 4, dtype=torch.int64, device="cuda")
    d_t = torch.empty(4, dtype=torch.int32, device="cuda")
    d_s = torch.empty(4, dtype=torch.int32, device="cuda")
    for strategy, method in (("lca*", "tree"), ("hybrid", "tree"), ("mrtl", "rmq")):
        with pytest.raises(api.UmgapError) as e:
            api.classify9_dev(ix, w.taxonomy, d_keys, 4, 84, d_t, d_s, method=method, strategy=strategy)
        assert e.value.code == -1 and "9-mer index" in e.value.message


# ---------------------------------------------------------------------------------------------
# remaining device-pointer entry points of include/umgap_hip.h
# ---------------------------------------------------------------------------------------------
def test_lookup_prot9_dev_matches_windows(api, torch_cuda):
    """umgap_index_lookup_prot9_dev = every 9-residue window of every peptide (src/commands/prot2kmer2lca.rs:172-179)."""
    torch = torch_cuda
    rng = np.random.default_rng(31)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    kmers = sorted({rng.choice(letters, 9).tobytes() for _ in range(5000)})
    vals = rng.integers(1, 1 << 30, len(kmers)).astype(np.uint32)
    keys, ok = api.pack9_array(kmers)
    assert ok.all()
    ix = api.Index.build(keys, vals)
    d = dict(zip(kmers, vals.tolist()))
    peps = []
    for _ in range(3000):
        n = int(rng.integers(0, 70))
        p = bytearray(rng.choice(letters, n).tobytes())
        if n >= 9 and rng.random() < 0.7:
            s = int(rng.integers(0, n - 8))
            p[s:s + 9] = kmers[int(rng.integers(0, len(kmers)))]
        if n and rng.random() < 0.2:
            p[int(rng.integers(0, n))] = ord("*")
        peps.append(bytes(p))
        if rng.random() < 0.01:
            peps += [b""] * int(rng.integers(60, 300)) + [b"ACDEFG"] * 5     # long runs of peptides without a window (window_locate.hpp's long way)
    data, offs = api._concat(peps)
    out_offs = np.zeros(len(peps) + 1, np.uint64)
    out_offs[1:] = np.cumsum([max(0, len(p) - 8) for p in peps])
    d_bytes = torch.from_numpy(data.copy()).cuda()
    d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    d_oo = torch.from_numpy(out_offs.view(np.int64)).cuda()
    d_out = torch.full((int(out_offs[-1]),), -1, dtype=torch.int32, device="cuda")
    ix.lookup_prot9_dev(d_bytes, d_offs, d_oo, d_out, miss_value=0)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(np.uint32)
    want = np.array([d.get(p[i:i + 9], 0) for p in peps for i in range(max(0, len(p) - 8))], np.uint32)
    assert (got == want).all() and (want != 0).sum() > 1000


def test_device_pointer_variants(api, torch_cuda):
    """build9_dev, aggregate_dev (CSR on the device), set_shard: same answers as the host-pointer forms."""
    torch = torch_cuda
    from umgap_amd.synth import synthetic_taxonomy
    rng = np.random.default_rng(41)
    keys = np.unique(rng.integers(0, 1 << 45, 100_000, dtype=np.uint64))
    vals = rng.integers(1, 1 << 30, len(keys)).astype(np.uint32)
    ix_h = api.Index.build(keys, vals)
    ix_d = api.Index.build_dev(torch.from_numpy(keys.view(np.int64)).cuda(), torch.from_numpy(vals.view(np.int32)).cuda())
    q = np.concatenate([keys[:5000], rng.integers(0, 1 << 45, 5000, dtype=np.uint64)])
    assert (ix_h.lookup9(q) == ix_d.lookup9(q)).all()
    # set_shard: the union of the shards answers everything exactly once
    want = ix_h.lookup9(q)
    acc = np.zeros(len(q), np.uint32)
    for s in range(4):
        ix_d.set_shard(s, 4)
        r = ix_d.lookup9(q)
        assert ((acc != 0) & (r != 0)).sum() == 0
        acc |= r
    assert (acc == want).all()
    ids, par, valid, rank = synthetic_taxonomy(3000, 5000, max_depth=25, seed=2)
    tx = api.Taxonomy(ids, par, valid, rank)
    reads = []
    for _ in range(30):
        pool = rng.choice(ids, 6)
        reads += _random_reads(rng, pool, 30, 40)
    taxa, offs = _csr(reads)
    d_taxa = torch.from_numpy(taxa.view(np.int32)).cuda()
    d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    for method, strategy in (("tree", "lca*"), ("rmq", "mrtl"), ("tree", "hybrid"), ("rmq", "hybrid")):
        t, s = tx.aggregate(taxa, offs, method=method, strategy=strategy, lower_bound=1.0)
        dt = torch.empty(len(reads), dtype=torch.int32, device="cuda")
        ds = torch.empty_like(dt)
        tx.aggregate_dev(d_taxa, d_offs, dt, ds, method=method, strategy=strategy, lower_bound=1.0)
        torch.cuda.synchronize()
        assert (dt.cpu().numpy().view(np.uint32) == t).all() and (ds.cpu().numpy().view(np.uint32) == s).all()


def _run_ranks(world, fn):
    """Drives `fn(rank)` from one host thread per rank (the local communicator's exchange blocks until every rank has
    arrived); re-raises the first failure."""
    import threading
    errs = [None] * world

    def body(r):
        try:
            fn(r)
        except BaseException as e:  # noqa: BLE001
            errs[r] = e
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
        assert not t.is_alive(), "a rank is stuck in the exchange"
    for e in errs:
        if e is not None:
            raise e


@pytest.mark.parametrize("world,pieces,own_share", [(1, 1, "probed in place"), (2, 3, "probed in place"), (3, 8, "packed"), (8, 4, "probed in place"), (2, 2, "packed"),
                                                    (3, 5, "peer-written")])
def test_sharded_table_local_ranks_match_whole_table(api, torch_cuda, world, pieces, own_share, monkeypatch):
    """The N-rank sharded path of the C ABI (pack 6-byte keys -> exchange -> probe on the owner -> 3-byte values back ->
    gather + [seedextend] + aggregate), all ranks on this one GPU through the local communicator: every rank's answers are
    bit-identical to the whole-table kernels, for the plain get() and for the fused pipeline with every strategy."""
    torch = torch_cuda
    from umgap_amd.dist import Comm, Sharded
    from umgap_amd.synth import ADD_B, MUL_A, SyntheticWorkload
    # a rank's own keys are normally probed by the kernel that consumes the hits; UMGAP_SHARD_NO_LOCAL=1 (read when the group is
    # created) sends them through pack / probe / gather like everybody else's
    monkeypatch.setenv("UMGAP_SHARD_NO_LOCAL", "1" if own_share == "packed" else "0")
    # "peer-written": put mode ASKED FOR between thread-ranks that share the GPU -- refused (they share the process's hardware queues: a
    # stage spinning on a peer's counter can sit in front of the kernel that would raise it); the hub's copies serve
    monkeypatch.setenv("UMGAP_SHARD_PUT", "1" if own_share == "peer-written" else "0")
    monkeypatch.setenv("UMGAP_PUT_TIMEOUT_S", "20")
    n_tax, n_keys, stride = 20_000, 1_500_000, 84
    w = SyntheticWorkload(n_tax=n_tax, n_keys=n_keys)
    n_reads = [4001 + 173 * r for r in range(world)]                       # ragged: every rank owns a different number of reads
    max_keys = (max(n_reads) + pieces) // pieces * pieces * stride
    keys = [w.reads(10_000 * r, n_reads[r], kmers_per_pair=stride) for r in range(world)]
    want_hits = [w.index.lookup9_dev(k, miss_value=0).clone() for k in keys]
    modes = [("tree", "lca*", 0, 2.0), ("tree", "lca*", 2, 2.0), ("tree", "hybrid", 2, 1.0), ("rmq", "mrtl", 2, 1.0), ("rmq", "hybrid", 0, 0.0)]
    want = {}
    for r in range(world):
        for m in modes:
            t = torch.empty(n_reads[r], dtype=torch.int32, device="cuda")
            s = torch.empty_like(t)
            api.classify9_dev(w.index, w.taxonomy, keys[r], n_reads[r], stride, t, s, method=m[0], strategy=m[1], factor=0.25, lower_bound=m[3],
                              fragments_per_read=m[2], min_seed_size=3, max_gap_size=1)
            want[(r, m)] = (t, s)
    torch.cuda.synchronize()
    shards = [api.Index.synthetic(n_keys, w.d_taxon_of_row, MUL_A, ADD_B, 0.0, r, world) for r in range(world)]
    comms = Comm.local(world)
    got_hits = [torch.full_like(want_hits[r], -7) for r in range(world)]
    got = {}

    def rank_body(r):
        torch.cuda.set_device(0)
        sh = Sharded(shards[r], w.taxonomy, comms[r], max_keys, pieces=pieces)
        assert sh.value_bytes == 3 and sh.pieces == pieces
        if world > 1:
            assert sh.transport == "copy", sh.transport      # thread-ranks that share a device keep the hub's copies, asked or not
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(2):                                              # buffer sets are reused across calls
                sh.lookup9(keys[r], got_hits[r], miss_value=0, stream=st)
            for m in modes:
                t = torch.full((n_reads[r],), -1, dtype=torch.int32, device="cuda")
                s = torch.full_like(t, -1)
                sh.classify9(keys[r], n_reads[r], stride, t, s, method=m[0], strategy=m[1], factor=0.25, lower_bound=m[3], fragments_per_read=m[2],
                             min_seed_size=3, max_gap_size=1, stream=st)
                got[(r, m)] = (t, s)
        st.synchronize()
        sh.check()
        sh.close()

    _run_ranks(world, rank_body)
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(got_hits[r], want_hits[r]), (world, r)
        assert (want_hits[r] != 0).float().mean().item() > 0.5
        for m in modes:
            assert torch.equal(got[(r, m)][1], want[(r, m)][1]), (world, r, m, "status")
            assert torch.equal(got[(r, m)][0], want[(r, m)][0]), (world, r, m, "taxon")


@pytest.mark.parametrize("world,pieces,own_share,transport", [(3, 2, "probed in place", "put"), (2, 3, "packed", "put"), (2, 9, "probed in place", "put"),
                                                             (3, 2, "probed in place", "copy"), (2, 3, "packed", "copy")])
def test_sharded_table_ranks_as_processes(api, torch_cuda, world, pieces, own_share, transport):
    """The N-rank protocol with every rank its own PROCESS (own HIP context, own shard of the table, own reads): what
    `torchrun --nproc-per-node N` starts on a node, here with the ranks sharing the one GPU of the box.  RCCL refuses two ranks
    on one device, so the exchange runs over umgap_comm_init_ipc (peers' send buffers mapped with hipIpc handles, barrier in
    shared memory); pack / probe / gather are the shipped kernels.  Each rank checks its answers against the whole table
    (tests/ipc_rank_worker.py) -- hits and three fused pipelines -- and the communicator's own pattern test.
    transport "put" (the IPC communicator's default): the pack stores keys straight into the owners' receive buffers and the probe stores
    values straight into the requesters' gather buffers through mapped exchange blocks, per-piece counters on the device instead of
    copies and host barriers (9 pieces: the buffer sets and their counters go round more than twice per call); "copy"
    (UMGAP_SHARD_PUT=0): store and forward with host-synchronised copies."""
    import subprocess
    import sys
    import uuid
    here = os.path.dirname(os.path.abspath(__file__))
    name = "/umgap-test-%s" % uuid.uuid4().hex[:12]
    env = dict(os.environ, UMGAP_SHARD_NO_LOCAL="1" if own_share == "packed" else "0", UMGAP_IPC_TIMEOUT_S="240", UMGAP_SHARD_PUT="1" if transport == "put" else "0",
               UMGAP_PUT_TIMEOUT_S="20")
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "ipc_rank_worker.py"), name, str(world), str(r), str(pieces), transport], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0].decode())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d:\n%s" % (r, out[-3000:])
        assert "rank %d of %d" % (r, world) in out and "bit-identical" in out, out[-2000:]
    assert not os.path.exists("/dev/shm" + name)       # the control block's name is gone once everybody is attached


def test_bench_gpus_2_control_flow_rehearsed_on_one_gpu(api, torch_cuda):
    """`python bench.py --gpus 2 --rehearse-ipc`: the WHOLE --gpus N control flow of the benchmark -- self-launch of one process per rank,
    rendezvous on 127.0.0.1, ranks_seen by all-reduce, the replica leg with barriers and the max over ranks, then the sharded step in a
    child process per rank (own process group, shard r of N, exchange, bit-for-bit check against the replica's answers), one JSON line
    from rank 0 -- with both ranks on the ONE GPU of the box: gloo stands in for RCCL's process group (RCCL refuses two ranks on a
    device) and the hipIpc communicator with peer-written keys and values for RCCL's send / receive.  What a node adds to this path is
    ncclSend / ncclRecv (or the same peer mappings over xGMI) and one GPU per rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-ipc", "--steps", "3", "--warmup", "1", "--n-keys", "400000000", "--n-tax", "300000",
           "--pairs-per-step", "1000000", "--batches", "2", "--pieces", "4", "--launch-timeout", "400"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["UMGAP_PUT_TIMEOUT_S"] = "20"
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                     # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 3 and out["value"] > 0
    assert "REHEARSAL" in out["config"]["parallelism"] and out["config"]["parity"]["answers_bit_identical_to_unbound_table"] is True
    sh = out["sharded"]
    assert "error" not in sh, sh
    assert sh["ranks_seen"] == 2 and sh["bit_identical_to_replica"] is True and sh["value"] > 0 and sh["pieces"] == 4
    assert "exchange: put" in sh["transport"] and "peers' mapped buffers" in sh["exchange"]


def test_comm_ipc_rendezvous_failures(api, torch_cuda, monkeypatch):
    """umgap_comm_init_ipc when the other ranks never come, or the arguments are wrong: an error after UMGAP_IPC_TIMEOUT_S (not a hang),
    and no shared-memory block left behind."""
    import time
    import uuid
    from umgap_amd.dist import Comm
    monkeypatch.setenv("UMGAP_IPC_TIMEOUT_S", "2")
    name = "/umgap-test-%s" % uuid.uuid4().hex[:12]
    t0 = time.perf_counter()
    with pytest.raises(api.UmgapError, match="waited"):
        Comm.ipc(name, 2, 0, 0)                      # rank 1 never shows up
    assert time.perf_counter() - t0 < 30 and not os.path.exists("/dev/shm" + name)
    with pytest.raises(api.UmgapError, match="did not create"):
        Comm.ipc(name, 2, 1, 0)                      # ... nor does rank 0
    for bad in ("no-slash", "/a/b", "/"):
        with pytest.raises(api.UmgapError, match="must look like"):
            Comm.ipc(bad, 1, 0, 0)
    with pytest.raises(api.UmgapError, match="rank < world"):
        Comm.ipc(name, 2, 2, 0)
    c = Comm.ipc(name, 1, 0, 0)                      # a world of one: nothing to wait for
    c.selftest(1 << 16)
    c.close()
    assert not os.path.exists("/dev/shm" + name)


@pytest.mark.parametrize("world,stride,frags,pieces,own_share", [(3, 128, 0, 2, "probed in place"), (2, 40, 2, 3, "probed in place"), (4, 7, 0, 1, "probed in place"),
                                                                 (2, 64, 64, 2, "probed in place"), (5, 1, 0, 2, "probed in place"), (2, 33, 33, 1, "probed in place"),
                                                                 (1, 84, 2, 2, "packed"), (3, 84, 0, 2, "packed"), (2, 40, 2, 3, "packed")])
def test_sharded_runs_other_strides_and_deferred_reads(api, torch_cuda, world, stride, frags, pieces, own_share, monkeypatch):
    """The exchange stores no per-key slot: pack and consumer walk the same RUNS of reads (the consumer's group: it depends on the
    k-mers and fragments per read) and recompute every key's slot.  Other run geometries than the benchmark's, reads with more
    than 64 distinct taxa (the large-scratch pass finds a deferred read inside its run), keys that are not packed 9-mers, and
    an empty rank."""
    torch = torch_cuda
    from umgap_amd.api import Index, Taxonomy
    from umgap_amd.dist import Comm, Sharded
    from umgap_amd.synth import synthetic_taxonomy
    # "packed" (UMGAP_SHARD_NO_LOCAL=1): no key is the rank's own -- and a key that is no packed 9-mer has NO owner either, which
    # must not pass for "own" (it was probed with its high bits masked off, out of the shard's bounds at world > 1)
    monkeypatch.setenv("UMGAP_SHARD_NO_LOCAL", "1" if own_share == "packed" else "0")
    ids, par, valid, rank = synthetic_taxonomy(5000, 8000, max_depth=12, seed=9)
    tx = Taxonomy(ids, par, valid, rank)
    rng = np.random.default_rng(1000 * world + stride)
    n_keys = 50_000
    keys = rng.permutation(np.unique(rng.integers(0, 1 << 45, 2 * n_keys, dtype=np.uint64)))[:n_keys]
    vals = rng.choice(ids, n_keys).astype(np.uint32)
    whole = Index.build(keys, vals)
    n_reads = [0 if (r == 1 and world > 2) else 700 + 37 * r for r in range(world)]     # one rank has nothing to classify
    qs = []
    for r in range(world):
        q = keys[rng.integers(0, n_keys, n_reads[r] * stride)]                         # all hits: as many distinct taxa as k-mers
        few = (n_reads[r] // 3) * stride
        q[:few] = keys[rng.integers(0, 40, few)]                                         # a third of the reads: few distinct taxa
        miss = rng.random(len(q)) < 0.2
        q[miss] = rng.integers(0, 1 << 45, int(miss.sum()), dtype=np.uint64)
        q[rng.random(len(q)) < 0.01] = np.uint64((1 << 45) + 12345)                      # not a packed 9-mer: never sent, a miss
        bad = rng.random(len(q)) < 0.01                                                  # ... nor is a PRESENT key with a bit above 45 set:
        q[bad] = keys[rng.integers(0, n_keys, int(bad.sum()))] | np.uint64(1 << 45)      # masked to 45 bits it would hit
        qs.append(torch.from_numpy(q.view(np.int64)).cuda())
    max_keys = (max(n_reads) + pieces) // pieces * pieces * stride
    modes = [("tree", "lca*", 0.0), ("tree", "hybrid", 0.0), ("rmq", "mrtl", 1.0), ("rmq", "hybrid", 0.0)]
    want, want_hits = {}, []
    for r in range(world):
        want_hits.append(whole.lookup9_dev(qs[r], miss_value=0).clone() if n_reads[r] else torch.empty(0, dtype=torch.int32, device="cuda"))
        for m in modes:
            t = torch.empty(n_reads[r], dtype=torch.int32, device="cuda")
            s = torch.empty_like(t)
            if n_reads[r]:
                api.classify9_dev(whole, tx, qs[r], n_reads[r], stride, t, s, method=m[0], strategy=m[1], factor=0.3, lower_bound=m[2],
                                  fragments_per_read=frags, min_seed_size=2, max_gap_size=1)
            want[(r, m)] = (t, s)
    torch.cuda.synchronize()
    if stride >= 100:
        h = want_hits[0].cpu().numpy().view(np.uint32)
        assert max(len(set(h[i * stride:(i + 1) * stride].tolist())) for i in range(n_reads[0])) > 64   # the deferred pass is exercised
    shards = []
    for r in range(world):
        ix = Index.build(keys, vals)
        ix.set_shard(r, world)
        shards.append(ix)
    comms = Comm.local(world)
    got, got_hits = {}, [torch.full_like(want_hits[r], -7) for r in range(world)]

    def rank_body(r):
        torch.cuda.set_device(0)
        sh = Sharded(shards[r], tx, comms[r], max_keys, pieces=pieces)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            sh.lookup9(qs[r], got_hits[r], miss_value=0, stream=st)
            for m in modes:
                t = torch.full((n_reads[r],), -1, dtype=torch.int32, device="cuda")
                s = torch.full_like(t, -1)
                sh.classify9(qs[r], n_reads[r], stride, t, s, method=m[0], strategy=m[1], factor=0.3, lower_bound=m[2], fragments_per_read=frags,
                             min_seed_size=2, max_gap_size=1, stream=st)
                got[(r, m)] = (t, s)
        st.synchronize()
        sh.check()
        sh.close()

    _run_ranks(world, rank_body)
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(got_hits[r], want_hits[r]), (world, r)
        for m in modes:
            assert torch.equal(got[(r, m)][1], want[(r, m)][1]), (world, r, m, "status")
            assert torch.equal(got[(r, m)][0], want[(r, m)][0]), (world, r, m, "taxon")


def test_sharded_table_wide_values_and_overflow(api, torch_cuda):
    """Values that need 4 bytes switch the return format; a region that overflows is reported by check() (sticky over
    several calls) and its keys are answered with a miss, never with another key's value."""
    torch = torch_cuda
    from umgap_amd.dist import Comm, Sharded
    rng = np.random.default_rng(77)
    keys = np.unique(rng.integers(0, 1 << 45, 200_000, dtype=np.uint64))
    vals = rng.integers(1 << 24, 1 << 32, len(keys), dtype=np.uint64).astype(np.uint32)
    vals[vals == 0xFFFFFFFF] = 5
    world = 2
    whole = api.Index.build(keys, vals)
    q = np.concatenate([keys[:60_000], rng.integers(0, 1 << 45, 20_000, dtype=np.uint64), np.array([1 << 45, (1 << 64) - 1], np.uint64)])
    d_q = torch.from_numpy(q.view(np.int64)).cuda()
    want = whole.lookup9_dev(d_q, miss_value=9).clone()
    shards = []
    for r in range(world):
        ix = api.Index.build(keys, vals)
        ix.set_shard(r, world)
        shards.append(ix)
    comms = Comm.local(world)
    outs = [torch.empty(len(q), dtype=torch.int32, device="cuda") for _ in range(world)]
    tripped = [False] * world
    # a present key owned by each rank (a rank's OWN keys never travel, so only somebody else's key can overflow a region)
    owner = whole.owner9_dev(d_q[:2000], world).cpu().numpy()
    key_of_owner = [int(np.nonzero(owner == o)[0][0]) for o in range(world)]

    def rank_body(r):
        torch.cuda.set_device(0)
        sh = Sharded(shards[r], None, comms[r], len(q), pieces=2)
        assert sh.value_bytes == 4
        st = torch.cuda.Stream()
        sh.lookup9(d_q, outs[r], miss_value=9, stream=st)
        sh.check()
        # a batch that goes to ONE owner overflows its region: flagged, and still flagged after a clean call
        other = key_of_owner[(r + 1) % world]
        skew = d_q[other:other + 1].repeat(len(q))
        o2 = torch.empty(len(q), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        sh.lookup9(skew, o2, miss_value=9, stream=st)
        sh.lookup9(d_q, outs[r], miss_value=9, stream=st)
        try:
            sh.check()
        except api.UmgapError as e:
            tripped[r] = "overflowed" in e.message
        v = o2.cpu().numpy().view(np.uint32)
        assert set(np.unique(v).tolist()) <= {9, int(want[other].item()) & 0xFFFFFFFF}  # the right value or a miss, nothing else
        sh.check()                                                                      # the flag is cleared by the report
        sh.close()

    _run_ranks(world, rank_body)
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(outs[r], want)
    assert all(tripped)


def test_sharded_table_rccl_world1(api, torch_cuda):
    """The RCCL transport (ncclSend/ncclRecv groups resolved with dlopen) on a one-rank communicator: a self-exchange,
    which is all a one-GPU box can run; the N-rank protocol itself is covered by the local ranks above."""
    torch = torch_cuda
    from umgap_amd.dist import Comm, Sharded
    from umgap_amd.synth import SyntheticWorkload
    w = SyntheticWorkload(n_tax=5000, n_keys=600_000)
    n_reads, stride = 5001, 84
    keys = w.reads(0, n_reads)
    want_hits = w.index.lookup9_dev(keys).clone()
    t0 = torch.empty(n_reads, dtype=torch.int32, device="cuda")
    s0 = torch.empty_like(t0)
    api.classify9_dev(w.index, w.taxonomy, keys, n_reads, stride, t0, s0, method="tree", lower_bound=2.0, fragments_per_read=2, min_seed_size=3, max_gap_size=1)
    comm = Comm.init_rank(Comm.unique_id(), 1, 0, 0)
    assert comm.world == 1 and comm.rank == 0
    sh = Sharded(w.index, w.taxonomy, comm, (n_reads + 4) // 5 * 5 * stride, pieces=5)
    hits = torch.full_like(want_hits, -7)
    t1, s1 = torch.full_like(t0, -1), torch.full_like(s0, -1)
    for _ in range(3):
        sh.lookup9(keys, hits)
        sh.classify9(keys, n_reads, stride, t1, s1, method="tree", lower_bound=2.0, fragments_per_read=2, min_seed_size=3, max_gap_size=1)
    sh.check()
    assert torch.equal(hits, want_hits) and torch.equal(t1, t0) and torch.equal(s1, s0)
    assert sh.probe_ms() > 0
    sh.close()
    comm.close()


def test_comm_selftest_drives_the_transport(api, torch_cuda):
    """umgap_comm_selftest moves a known pattern to every rank INCLUDING the sender through the lookup's own transport:
    on the one-rank RCCL communicator that is a real ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on a side stream
    (the calls the N-rank exchange makes and a one-GPU box cannot otherwise reach); on the local hub, three ranks."""
    from umgap_amd.dist import Comm
    comm = Comm.init_rank(Comm.unique_id(), 1, 0, 0)
    for nbytes in (1, 4099, 3 << 20):
        comm.selftest(nbytes)
    comm.close()
    comms = Comm.local(3)
    _run_ranks(3, lambda r: (torch_cuda.cuda.set_device(0), comms[r].selftest(70001)))
    for c in comms:
        c.close()


# ---------------------------------------------------------------------------------------------
# surface B: protein fragments in, fused kernel (LDS-staged k-mer packing) vs the oracle stages
# ---------------------------------------------------------------------------------------------
def _proteome(torch, w, n_frag, flen=50, seed=7):
    from umgap_amd import _lib
    from umgap_amd.api import _dev_ptr, _stream_ptr
    offs = torch.arange(0, (n_frag + 1) * flen, flen, dtype=torch.int64, device="cuda")
    data = torch.empty(n_frag * flen, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.load().umgap_synth_peptides_dev(0, n_frag, seed, _dev_ptr(offs), _dev_ptr(data), 0, _stream_ptr(None)))
    taxa = w.d_taxon_of_row[torch.arange(n_frag, device="cuda") % w.n_tax].contiguous()
    return data, offs, taxa


def _pack_windows(arr2d):
    """uint8 [n, L] -> (keys uint64 [n, L-8], ok bool [n, L-8])"""
    n, L = arr2d.shape
    code = arr2d.astype(np.int64) - 65
    good = (code >= 0) & (code < 26)
    keys = np.zeros((n, L - 8), np.uint64)
    ok = np.ones((n, L - 8), bool)
    for c in range(9):
        keys = (keys << np.uint64(5)) | (code[:, c:c + L - 8] & 31).astype(np.uint64)
        ok &= good[:, c:c + L - 8]
    return keys, ok


@pytest.mark.parametrize("mode", [("tree", "lca*"), ("tree", "hybrid"), ("rmq", "hybrid"), ("rmq", "mrtl")])
@pytest.mark.parametrize("with_seedextend", [False, True])
@pytest.mark.parametrize("F", [2, 12])
def test_fused_classify_prot9_vs_oracle(api, torch_cuda, with_seedextend, mode, F):
    """umgap_classify_prot9[_strategy]_dev (fragments in, one taxon per read out) against the oracle's stages, every strategy.
    F = 12: a pair translated in all six frames (scripts/umgap-analyse.sh:277-288) -- 504 windows per read, the LONG variant of the
    kernel; two of the twelve fragments come from the proteome (the frames that code), the others are random residues."""
    torch = torch_cuda
    method, strategy = mode
    import ctypes as C
    from umgap_amd import _lib
    from umgap_amd.api import Index, _dev_ptr, _stream_ptr
    from umgap_amd.synth import SyntheticWorkload
    L = _lib.load()
    n_tax, n_frag, flen, n_reads = 5000, 20_000, 50, 6000 if F == 2 else 3001
    w = SyntheticWorkload(n_tax=n_tax, n_keys=1000, build_index=False)
    prot, poffs, ptaxa = _proteome(torch, w, n_frag, flen)
    h = C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(poffs), _dev_ptr(ptaxa), n_frag, flen, n_frag * (flen - 8), 0.0, 0, C.byref(h)))
    ix = Index(h.value)
    reads = torch.empty(n_reads * 2 * flen, dtype=torch.uint8, device="cuda")
    _lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 44, 0, n_reads, 2, 30, _dev_ptr(reads), 0, _stream_ptr(None)))
    if F != 2:     # the two coding fragments become frames 0 and F / 2 of the read; the other frames are random residues
        letters = torch.tensor(list(b"ACDEFGHIKLMNPQRSTVWY"), dtype=torch.uint8, device="cuda")
        g = torch.Generator(device="cuda").manual_seed(5)
        frames = letters[torch.randint(0, 20, (n_reads, F, flen), device="cuda", generator=g)]
        frames[:, 0, :] = reads.view(n_reads, 2, flen)[:, 0, :]
        frames[:, F // 2, :] = reads.view(n_reads, 2, flen)[:, 1, :]
        reads = frames.reshape(-1).contiguous()
    roffs = torch.arange(0, (n_reads * F + 1) * flen, flen, dtype=torch.int64, device="cuda")
    d_t = torch.empty(n_reads, dtype=torch.int32, device="cuda")
    d_s = torch.empty_like(d_t)
    api.classify_prot9_dev(ix, w.taxonomy, reads, roffs, n_reads, F, d_t, d_s, method=method, strategy=strategy, factor=0.25, lower_bound=2.0,
                           seedextend=with_seedextend, min_seed_size=3, max_gap_size=1)
    torch.cuda.synchronize()
    # oracle side: the index as a sorted (key -> min taxon) array, lookups by searchsorted
    pk, pok = _pack_windows(prot.cpu().numpy().reshape(n_frag, flen))
    pt = np.repeat(ptaxa.cpu().numpy().view(np.uint32), flen - 8)
    order = np.lexsort((pt, pk.ravel()))
    ks, ts = pk.ravel()[order], pt[order]
    first = np.concatenate([[True], ks[1:] != ks[:-1]])
    ukeys, utax = ks[first], ts[first]                       # smallest taxon per distinct 9-mer
    assert pok.all()
    rk, rok = _pack_windows(reads.cpu().numpy().reshape(n_reads * F, flen))
    pos = np.searchsorted(ukeys, rk.ravel())
    pos[pos >= len(ukeys)] = 0
    hits = np.where((ukeys[pos] == rk.ravel()) & rok.ravel(), utax[pos], 0).astype(np.uint32)
    assert (0.3 if F == 2 else 0.05) < (hits != 0).mean() < 0.95
    # the unfused device path for the same bytes gives the same hits
    oo = torch.arange(0, (n_reads * F + 1) * (flen - 8), flen - 8, dtype=torch.int64, device="cuda")
    d_hits = torch.empty(n_reads * F * (flen - 8), dtype=torch.int32, device="cuda")
    ix.lookup_prot9_dev(reads, roffs, oo, d_hits)
    torch.cuda.synchronize()
    assert (d_hits.cpu().numpy().view(np.uint32) == hits).all()
    per = flen - 8
    if with_seedextend:
        out = np.zeros_like(hits)
        for r in range(n_reads * F):
            kept = O.seedextend_record(hits[r * per:(r + 1) * per], 3, 1)
            out[r * per:r * per + len(kept)] = kept
        hits = out
    agg = O.Aggregator(O.Taxonomy.from_arrays(w.ids, w.parents, w.valid, w.rank), method, strategy, 0.25)
    offs = np.arange(n_reads + 1, dtype=np.uint64) * (F * per)
    t, s = d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32)
    nbad, bad, wt, ws = agg.taxa2agg_check(hits, offs, t, s, lower_bound=2.0, nthreads=8)
    assert nbad == 0, (bad, t[bad.astype(int)], s[bad.astype(int)], wt[bad.astype(int)])
    assert (s == 0).mean() > 0.5
    # the same launch on the table bound to the taxonomy: bit-identical
    ix.bind(w.taxonomy)
    d_t2, d_s2 = torch.empty_like(d_t), torch.empty_like(d_s)
    api.classify_prot9_dev(ix, w.taxonomy, reads, roffs, n_reads, F, d_t2, d_s2, method=method, strategy=strategy, factor=0.25, lower_bound=2.0,
                           seedextend=with_seedextend, min_seed_size=3, max_gap_size=1)
    torch.cuda.synchronize()
    ix.unbind(w.taxonomy)
    assert torch.equal(d_t, d_t2) and torch.equal(d_s, d_s2)


@pytest.mark.parametrize("mutations_per_mille,shift", [(0, 0), (30, 0), (250, 0), (0, 1), (30, 1), (30, 2), (250, 2)])
def test_walk_index_matches_plain_table(api, torch_cuda, mutations_per_mille, shift, monkeypatch):
    """The 9-mer index in walk order (umgap_index_build_walk9_dev, index_walk.hip): same proteome, same values as the plain table
    (umgap_index_build_windows9_dev: smallest taxon on duplicate 9-mers), so every answer must be bit-identical -- get() of packed
    9-mers, the windows of protein fragments, and the fused surface-B pipeline for every strategy, with and without seedextend --
    on reads that follow the proteome exactly, that deviate from it (every break re-anchors the walk), and that hardly match at
    all.  The proteome has fragments of every length (shorter than 9, with '*' and lower-case bytes inside) and whole fragments
    repeated under another taxon (duplicate 9-mers: the walk position is the first occurrence, the value the smallest taxon)."""
    torch = torch_cuda
    import ctypes as C
    from umgap_amd import _lib
    from umgap_amd.api import Index, _dev_ptr, _stream_ptr
    from umgap_amd.synth import SyntheticWorkload
    L = _lib.load()
    # shift > 0: the wide form of the index (round 4; what 8 G keys and more need) forced on this small one -- the table holds the entry
    # number >> shift, a fragment's run of entries starts at a multiple of 2^shift, and the key's low bits pick the entry of the group
    # (walk_ref.hpp); the low-complexity fragments below put keys with EQUAL low bits into one group (the exception list)
    monkeypatch.setenv("UMGAP_WALK_SHIFT", str(shift))
    rng = np.random.default_rng(5 + mutations_per_mille)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    w = SyntheticWorkload(n_tax=4000, n_keys=1000, build_index=False)
    # proteome: ragged fragments, some odd bytes, some repeated
    frags = []
    for i in range(30_000):
        n = int(rng.choice([0, 3, 8, 9, 10, 25, 50, 50, 50, 73, 120]))
        f = bytearray(rng.choice(letters, n).tobytes())
        if n >= 25 and rng.random() < 0.03:     # a homopolymer stretch: consecutive 9-mers that differ in their first residues only
            a = int(rng.integers(0, n - 20))
            f[a:a + 18] = bytes([int(rng.choice(letters))]) * 18
        if n > 12 and rng.random() < 0.05:
            f[int(rng.integers(0, n))] = int(rng.choice([42, 88 + 32, 64]))      # '*', 'x', '@'
        frags.append(bytes(f))
    frags += [frags[int(i)] for i in rng.integers(0, len(frags), 3000)]            # duplicates, other taxa below
    n_frag = len(frags)
    pdata = np.frombuffer(b"".join(frags), np.uint8)
    poffs = np.zeros(n_frag + 1, np.int64)
    poffs[1:] = np.cumsum([len(f) for f in frags])
    ptaxa = rng.choice(w.ids, n_frag).astype(np.uint32)
    d_p = torch.from_numpy(pdata.copy()).cuda()
    d_po = torch.from_numpy(poffs).cuda()
    d_pt = torch.from_numpy(ptaxa.view(np.int32)).cuda()
    max_len = max(len(f) for f in frags)
    n_win = int(sum(max(0, len(f) - 8) for f in frags))
    hp, hw = C.c_void_p(), C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(d_p), _dev_ptr(d_po), _dev_ptr(d_pt), n_frag, max_len, n_win, 0.0, 0, C.byref(hp)))
    _lib.check(L.umgap_index_build_walk9_dev(_dev_ptr(d_p), _dev_ptr(d_po), _dev_ptr(d_pt), n_frag, max_len, n_win, 0.0, 0, C.byref(hw)))
    plain, walk = Index(hp.value), Index(hw.value)
    assert plain.kind == 9 and walk.kind == 10 and walk.num_keys == plain.num_keys
    w_shift, w_entries, w_exc = walk.walk_info()
    assert w_shift == shift and w_entries > 0
    assert (w_exc > 0) == (shift > 0), w_exc       # the homopolymer stretches: keys with equal low bits side by side
    # reads: mutated copies of the 50-residue fragments, plus ragged reads cut from the proteome and random ones
    # (lengths 73 and 120: runs of up to 16 windows per lane with room behind a substituted residue for the walk to pick up again)
    long_ids = np.array([i for i, f in enumerate(frags) if len(f) >= 50 and all(65 <= c <= 90 for c in f)], np.int64)
    n_reads, F = 3000, 2
    rfr = []
    for r in range(n_reads * F):
        c = rng.random()
        if c < 0.7:
            f = bytearray(frags[int(rng.choice(long_ids))])
            for k in range(len(f)):
                if rng.random() * 1000 < mutations_per_mille:
                    f[k] = int(rng.choice(letters))
            if mutations_per_mille and len(f) > 20:
                e = rng.random()
                k = int(rng.integers(1, len(f) - 1))
                if e < 0.12:
                    del f[k]                                                   # a deletion: the walk's offset no longer fits behind it
                elif e < 0.24:
                    f.insert(k, int(rng.choice(letters)))                      # an insertion
                elif e < 0.34:
                    g = frags[int(rng.choice(long_ids))]                       # a chimera: the read goes on in another protein
                    f = f[:k] + bytearray(g[min(k, len(g)):])
                elif e < 0.40:
                    f[k] = 42                                                  # '*': nine windows without a key
            a, b = (0, len(f)) if rng.random() < 0.6 else sorted(rng.integers(0, len(f) + 1, 2).tolist())
            rfr.append(bytes(f[a:b]))
        elif c < 0.85:
            src = frags[int(rng.integers(0, n_frag))]
            rfr.append(src[:int(rng.integers(0, min(len(src), 60) + 1))])
        else:
            rfr.append(rng.choice(letters, int(rng.integers(0, 40))).tobytes())
    rdata = np.frombuffer(b"".join(rfr) + b"A", np.uint8)[:-1]
    roffs = np.zeros(len(rfr) + 1, np.int64)
    roffs[1:] = np.cumsum([len(f) for f in rfr])
    d_r = torch.from_numpy(rdata.copy()).cuda() if len(rdata) else torch.empty(0, dtype=torch.uint8, device="cuda")
    d_ro = torch.from_numpy(roffs).cuda()
    # windows of the reads through both tables
    ooffs = np.zeros(len(rfr) + 1, np.int64)
    ooffs[1:] = np.cumsum([max(0, len(f) - 8) for f in rfr])
    d_oo = torch.from_numpy(ooffs).cuda()
    h0 = torch.full((int(ooffs[-1]),), -3, dtype=torch.int32, device="cuda")
    h1 = torch.full_like(h0, -5)
    plain.lookup_prot9_dev(d_r, d_ro, d_oo, h0, miss_value=0)
    walk.lookup_prot9_dev(d_r, d_ro, d_oo, h1, miss_value=0)
    torch.cuda.synchronize()
    assert torch.equal(h0, h1)
    if mutations_per_mille == 0:
        assert (h0 != 0).float().mean().item() > 0.5
    # packed 9-mers
    keys = np.unique(rng.integers(0, 1 << 45, 5000, dtype=np.uint64))
    pk = []
    for f in frags[:400]:
        for i in range(max(0, len(f) - 8)):
            k, ok = 0, True
            for c in f[i:i + 9]:
                ok = ok and 65 <= c <= 90
                k = (k << 5) | ((c - 65) & 31)
            if ok:
                pk.append(k)
    q = torch.from_numpy(np.concatenate([keys, np.array(pk, np.uint64)]).view(np.int64)).cuda()
    assert torch.equal(plain.lookup9_dev(q, miss_value=7), walk.lookup9_dev(q, miss_value=7))
    # the fused pipeline
    for method, strategy, lb, se in (("tree", "lca*", 2.0, False), ("tree", "lca*", 1.0, True), ("rmq", "lca*", 0.0, False), ("tree", "hybrid", 1.0, True),
                                     ("rmq", "mrtl", 1.0, False), ("rmq", "hybrid", 0.0, True)):
        outs = []
        for ix in (plain, walk):
            t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
            st = torch.full_like(t, -1)
            api.classify_prot9_dev(ix, w.taxonomy, d_r, d_ro, n_reads, F, t, st, method=method, strategy=strategy, factor=0.25, lower_bound=lb,
                                   seedextend=se, min_seed_size=2, max_gap_size=1)
            outs.append((t, st))
        torch.cuda.synchronize()
        assert torch.equal(outs[0][1], outs[1][1]), (method, strategy, "status")
        assert torch.equal(outs[0][0], outs[1][0]), (method, strategy, "taxon")
    # surface A on the walk index (round 5): PACKED 9-mers whose neighbours are consecutive windows -- what pept2lca gets from prot2kmer --
    # through umgap_classify9[_strategy]_dev: one probe per run, the rest from the walk array; answers those of the plain table.  Reads
    # of 2 x 42 windows (mutated proteome fragments, chimeras, '*' inside: keys that are no 9-mer), with and without the fragment
    # structure (without it a run may cross from one fragment's last window into the next one's first), other strides
    frag50 = [frags[int(i)][:50] for i in long_ids]
    n2 = 1500
    seqs = []
    for r in range(n2 * 2):
        f = bytearray(frag50[int(rng.integers(0, len(frag50)))])
        for k in range(50):
            if rng.random() * 1000 < mutations_per_mille:
                f[k] = int(rng.choice(letters))
        if mutations_per_mille and rng.random() < 0.1:
            f[int(rng.integers(0, 50))] = 42
        if mutations_per_mille and rng.random() < 0.1:
            g = frag50[int(rng.integers(0, len(frag50)))]
            k = int(rng.integers(1, 49))
            f = f[:k] + bytearray(g[k:])
        seqs.append(bytes(f))
    arr = np.frombuffer(b"".join(seqs), np.uint8).reshape(n2 * 2, 50)
    k2, ok2 = _pack_windows(arr)
    k2 = np.where(ok2, k2, np.uint64(1 << 50)).reshape(-1)                     # (not a packed 9-mer: a miss, and the end of a walk)
    d_k2 = torch.from_numpy(k2.view(np.int64).copy()).cuda()
    for stride, fr, method, strategy, lb, se in ((84, 2, "tree", "lca*", 2.0, True), (84, 2, "tree", "lca*", 1.0, False), (84, 0, "rmq", "lca*", 0.0, False),
                                                 (42, 1, "tree", "hybrid", 1.0, True), (84, 2, "rmq", "mrtl", 1.0, True), (28, 0, "rmq", "hybrid", 0.0, False), (126, 3, "tree", "lca*", 2.0, True)):
        nr = len(k2) // stride
        outs = []
        for ix in (plain, walk):
            t = torch.full((nr,), -1, dtype=torch.int32, device="cuda")
            st = torch.full_like(t, -1)
            api.classify9_dev(ix, w.taxonomy, d_k2, nr, stride, t, st, method=method, strategy=strategy, factor=0.25, lower_bound=lb,
                              fragments_per_read=fr if se or fr else 0, min_seed_size=2 if se else 1, max_gap_size=1)
            outs.append((t, st))
        torch.cuda.synchronize()
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0]), (stride, fr, method, strategy)
        if mutations_per_mille == 0 and lb < 2:
            assert (outs[0][1] == 0).float().mean().item() > 0.5
    # bound to the taxonomy (the walk array's taxa become interval codes, the table of window numbers stays): same answers, and
    # the plain get() calls are refused while bound, as for the plain table
    walk.bind(w.taxonomy)
    for method, strategy, lb, se in (("tree", "lca*", 2.0, False), ("rmq", "mrtl", 1.0, True), ("tree", "hybrid", 0.0, False)):
        t0 = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
        s0 = torch.full_like(t0, -1)
        t1, s1 = torch.full_like(t0, -2), torch.full_like(t0, -2)
        api.classify_prot9_dev(plain, w.taxonomy, d_r, d_ro, n_reads, F, t0, s0, method=method, strategy=strategy, factor=0.25, lower_bound=lb, seedextend=se, min_seed_size=2, max_gap_size=1)
        api.classify_prot9_dev(walk, w.taxonomy, d_r, d_ro, n_reads, F, t1, s1, method=method, strategy=strategy, factor=0.25, lower_bound=lb, seedextend=se, min_seed_size=2, max_gap_size=1)
        torch.cuda.synchronize()
        assert torch.equal(t0, t1) and torch.equal(s0, s1), (method, strategy, "bound walk")
    with pytest.raises(api.UmgapError):
        walk.lookup9_dev(q)
    walk.unbind(w.taxonomy)
    assert torch.equal(plain.lookup9_dev(q, miss_value=7), walk.lookup9_dev(q, miss_value=7))
    # the file (UMGAPIX1 kind 10): saved, opened, and every entry point of the commands on it -- get() of host keys, the record
    # path of prot2kmer2agg, the dump behind printindex -- answers as the plain table does; so do the commands themselves
    import subprocess
    import tempfile
    from umgap_amd.api import classify_records
    with tempfile.TemporaryDirectory() as td:
        fp, fw = os.path.join(td, "plain.ix"), os.path.join(td, "walk.ix")
        plain.save(fp)
        walk.save(fw)
        n_entries = (os.path.getsize(fw) - os.path.getsize(fp)) // 8      # the walk array: one entry per distinct 9-mer + kept repeats
        assert 0 < n_entries <= n_win and (os.path.getsize(fw) - os.path.getsize(fp)) % 8 == 0
        walk2 = Index.open(fw)
        assert walk2.kind == 10 and walk2.num_keys == plain.num_keys
        with pytest.raises(api.UmgapError):
            Index.open(fw, shard=0, n_shards=2)
        assert torch.equal(plain.lookup9_dev(q, miss_value=7), walk2.lookup9_dev(q, miss_value=7))
        hq = q.cpu().numpy().view(np.uint64)
        assert np.array_equal(plain.lookup9(hq, miss_value=3), walk2.lookup9(hq, miss_value=3))
        t2, s2 = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda"), torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
        t3, s3 = torch.full_like(t2, -2), torch.full_like(t2, -2)
        api.classify_prot9_dev(plain, w.taxonomy, d_r, d_ro, n_reads, F, t2, s2, method="tree", strategy="lca*", lower_bound=2.0)
        api.classify_prot9_dev(walk2, w.taxonomy, d_r, d_ro, n_reads, F, t3, s3, method="tree", strategy="lca*", lower_bound=2.0)
        torch.cuda.synchronize()
        assert torch.equal(t2, t3) and torch.equal(s2, s3)
        rf = np.arange(0, len(rfr) + 1, F, dtype=np.uint64)
        for se in (False, True):
            a = classify_records(plain, w.taxonomy, rfr, rf, method="rmq", strategy="mrtl", lower_bound=1.0, seedextend=se, min_seed_size=2, max_gap_size=1)
            b = classify_records(walk2, w.taxonomy, rfr, rf, method="rmq", strategy="mrtl", lower_bound=1.0, seedextend=se, min_seed_size=2, max_gap_size=1)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        walk2.close()
        umgap = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "umgap_amd", "bin", "umgap")
        fasta = b"".join(b">r%d\n%s\n" % (i, f) for i, f in enumerate(rfr[:1500]))
        for args in (["prot2kmer2lca", "-o"], ["prot2kmer2lca"], ["printindex"]):
            outs = []
            for f in (fp, fw):
                r = subprocess.run([umgap] + args + [f], input=b"" if args[0] == "printindex" else fasta, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                assert r.returncode == 0, r.stderr
                outs.append(r.stdout)
            if args[0] == "printindex":
                assert sorted(outs[0].splitlines()) == sorted(outs[1].splitlines()) and len(outs[0].splitlines()) <= n_entries <= len(outs[0].splitlines()) * 9 // 8 + 1
            else:
                assert outs[0] == outs[1] and len(outs[0]) > 0, args


@pytest.mark.parametrize("n_chunks,load,shift,how", [(1, 0.0, 0, "table"), (3, 0.0, 0, "table"), (5, 0.9, 0, "table"), (3, 0.0, 1, "table"), (5, 0.9, 2, "table"),
                                                     (1, 0.0, 0, "deferred"), (3, 0.0, 1, "deferred"), (5, 0.9, 2, "deferred"), (4, 0.0, 0, "deferred")])
def test_walk_builder_values_from_a_table_order_from_a_proteome(api, torch_cuda, n_chunks, load, shift, how, monkeypatch):
    """umgap_walk_builder_*: the values come from a kind-9 table (what `umgap buildindex` makes from the reference's TSV), the order
    from a proteome handed over in chunks.  The table holds only part of the proteome's 9-mers (the others must miss) and keys the
    proteome never shows (appended as walks of one); fragments repeat within and across chunks (every key is listed once, at its
    first window).  Every answer -- get() of keys, the windows of reads, the fused pipeline -- is the values table's, bit for bit;
    the walk array has one entry per key, plus those of short runs of repeats inside a fragment (at most an eighth more; at the
    high load factor some keys go through the stash and may get a second).
    how = "deferred" (round 5, umgap_walk_builder_new_deferred): no values table on the device at all -- the proteome first (every
    9-mer of it gets an entry valued "no key"), then the (key, value) pairs in shuffled chunks; the keys the proteome never showed are
    appended; the same answers, a walk array of one entry per 9-mer of the proteome or key."""
    monkeypatch.setenv("UMGAP_WALK_SHIFT", str(shift))      # > 0: the wide entry numbers of walk_ref.hpp forced on this small index
    torch = torch_cuda
    from umgap_amd.api import Index
    from umgap_amd.synth import SyntheticWorkload
    rng = np.random.default_rng(100 + n_chunks)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    w = SyntheticWorkload(n_tax=3000, n_keys=1000, build_index=False)
    frags = []
    for i in range(12_000):
        n = int(rng.choice([0, 5, 9, 12, 30, 50, 50, 80, 200]))
        f = bytearray(rng.choice(letters[:int(rng.choice([2, 20, 20]))], n).tobytes())      # two-letter stretches: repeated 9-mers, equal low key bits
        if n > 12 and rng.random() < 0.05:
            f[int(rng.integers(0, n))] = int(rng.choice([42, 120]))
        frags.append(bytes(f))
    frags += [frags[int(i)] for i in rng.integers(0, len(frags), 4000)]
    order = rng.permutation(len(frags))
    frags = [frags[int(i)] for i in order]
    def keys_of(f):
        out = []
        for i in range(max(0, len(f) - 8)):
            k, ok = 0, True
            for c in f[i:i + 9]:
                ok = ok and 65 <= c <= 90
                k = (k << 5) | ((c - 65) & 31)
            if ok:
                out.append(k)
        return out
    prot_keys = np.unique(np.array([k for f in frags for k in keys_of(f)], np.uint64))
    keep = prot_keys[rng.random(len(prot_keys)) < 0.8]
    extra = np.setdiff1d(np.unique(rng.integers(0, 20 ** 9, 6000, dtype=np.uint64) * 7 % (1 << 45)), prot_keys)
    extra = extra[[all(((int(k) >> (5 * j)) & 31) < 26 for j in range(9)) for k in extra]]
    vkeys = np.concatenate([keep, extra])
    vtaxa = rng.choice(w.ids, len(vkeys)).astype(np.uint32)
    values = Index.build(vkeys, vtaxa, load_factor=load)
    cuts = np.linspace(0, len(frags), n_chunks + 1).astype(int)
    chunks, keepalive = [], []
    for c in range(n_chunks):
        part = frags[cuts[c]:cuts[c + 1]]
        data, offs = api._concat(part)
        d_b = torch.from_numpy(np.concatenate([data, np.zeros(1, np.uint8)])).cuda()
        d_o = torch.from_numpy(offs.view(np.int64).copy()).cuda()
        chunks.append((d_b, d_o, max(9, max((len(f) for f in part), default=9))))
    if how == "table":
        walk = Index.build_walk(values, chunks, load_factor=load)
    else:
        perm = rng.permutation(len(vkeys))
        vcuts = np.linspace(0, len(vkeys), 4).astype(int)
        vchunks = [(torch.from_numpy(vkeys[perm[a:b]].view(np.int64).copy()).cuda(), torch.from_numpy(vtaxa[perm[a:b]].view(np.int32).copy()).cuda()) for a, b in zip(vcuts[:-1], vcuts[1:])]
        vchunks.append(vchunks[0])                                   # (a chunk twice: the same values again are fine)
        walk = Index.build_walk_deferred(len(prot_keys) + 1000, chunks, vchunks, load_factor=load)
    assert walk.kind == 10 and walk.num_keys == len(vkeys)
    # get(): every key of the table, the proteome's keys the table does not hold, random keys
    q = np.concatenate([vkeys, np.setdiff1d(prot_keys, keep), np.unique(rng.integers(0, 1 << 45, 3000, dtype=np.uint64))])
    ref = dict(zip(vkeys.tolist(), vtaxa.tolist()))
    want = np.array([ref.get(int(k), 7) for k in q], np.uint32)
    d_q = torch.from_numpy(q.view(np.int64)).cuda()
    got_plain = values.lookup9_dev(d_q, miss_value=7).cpu().numpy().view(np.uint32)
    got_walk = walk.lookup9_dev(d_q, miss_value=7).cpu().numpy().view(np.uint32)
    assert np.array_equal(got_plain, want) and np.array_equal(got_walk, want)
    assert np.array_equal(walk.lookup9(q, miss_value=7), want)
    # the dump lists every key once with its taxon
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fw = os.path.join(td, "w.ix")
        walk.save(fw)
        with open(fw, "rb") as f:
            hdr = f.read(72)
        reserved = int(np.frombuffer(hdr[64:72], np.uint64)[0])
        n_walk = reserved & ((1 << 60) - 1)
        assert (reserved >> 60) & 3 == shift and (reserved >> 62) == (1 if how == "deferred" else 0)
        # (with a shift every fragment's run of entries is padded to a multiple of 2^shift; deferred: an entry per 9-mer of the proteome, key or not)
        n_ent = len(vkeys) if how == "table" else len(prot_keys) + len(extra)
        assert n_ent <= n_walk <= n_ent + n_ent // 8 + (0 if load < 0.8 else n_ent // 50) + ((1 << shift) - 1) * len(frags), (n_walk, n_ent)
        dk, dt = walk.dump9()
        assert len(dk) == len(vkeys) and dict(zip(dk.tolist(), dt.tolist())) == ref
        again = Index.open(fw)
        assert np.array_equal(again.lookup9(q, miss_value=7), want)
        again.close()
    # reads cut from the proteome (with a changed residue now and then): windows and the fused pipeline
    n_reads, F = 2500, 2
    rfr = []
    for r in range(n_reads * F):
        f = bytearray(frags[int(rng.integers(0, len(frags)))])
        if len(f) > 0 and rng.random() < 0.5:
            f[int(rng.integers(0, len(f)))] = int(rng.choice(letters))
        a, b = (0, len(f)) if rng.random() < 0.6 else sorted(rng.integers(0, len(f) + 1, 2).tolist())
        rfr.append(bytes(f[a:min(b, a + 60)]))
    rdata = np.frombuffer(b"".join(rfr) + b"A", np.uint8)[:-1]
    roffs = np.zeros(len(rfr) + 1, np.int64)
    roffs[1:] = np.cumsum([len(f) for f in rfr])
    d_r, d_ro = torch.from_numpy(rdata.copy()).cuda(), torch.from_numpy(roffs).cuda()
    ooffs = np.zeros(len(rfr) + 1, np.int64)
    ooffs[1:] = np.cumsum([max(0, len(f) - 8) for f in rfr])
    d_oo = torch.from_numpy(ooffs).cuda()
    h0 = torch.full((int(ooffs[-1]),), -3, dtype=torch.int32, device="cuda")
    h1 = torch.full_like(h0, -5)
    values.lookup_prot9_dev(d_r, d_ro, d_oo, h0, miss_value=0)
    walk.lookup_prot9_dev(d_r, d_ro, d_oo, h1, miss_value=0)
    torch.cuda.synchronize()
    assert torch.equal(h0, h1) and 0.2 < (h0 != 0).float().mean().item() < 0.95
    for method, strategy, lb, se in (("tree", "lca*", 2.0, False), ("rmq", "mrtl", 1.0, True), ("tree", "hybrid", 0.0, True)):
        outs = []
        for ix in (values, walk):
            t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
            st = torch.full_like(t, -1)
            api.classify_prot9_dev(ix, w.taxonomy, d_r, d_ro, n_reads, F, t, st, method=method, strategy=strategy, factor=0.25, lower_bound=lb,
                                   seedextend=se, min_seed_size=2, max_gap_size=1)
            outs.append((t, st))
        torch.cuda.synchronize()
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0]), (method, strategy)
    # a second builder over a bound or sharded values table is refused
    values.bind(w.taxonomy)
    with pytest.raises(api.UmgapError):
        Index.build_walk(values, chunks[:1])
    values.unbind(w.taxonomy)


def test_classify_prot9_ragged_fragments(api, torch_cuda):
    """Fragments of different lengths (incl. shorter than 9, empty, with '*'), 1..12 fragments per read."""
    torch = torch_cuda
    rng = np.random.default_rng(77)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    kmers = sorted({rng.choice(letters, 9).tobytes() for _ in range(3000)})
    ids = np.arange(1, 40, dtype=np.uint32)
    par = np.maximum(1, ids // 2).astype(np.uint32)
    par[0] = 1
    tx = api.Taxonomy(ids, par)
    otx = O.Taxonomy.from_arrays(ids, par)
    vals = rng.choice(ids, len(kmers)).astype(np.uint32)
    keys, _ = api.pack9_array(kmers)
    ix = api.Index.build(keys, vals)
    d = dict(zip(kmers, vals.tolist()))
    for F in (1, 2, 5, 12):
        frags = []
        for _ in range(400 * F):
            n = int(rng.choice([0, 3, 8, 9, 10, 17]))
            p = bytearray(rng.choice(letters, n).tobytes())
            if n >= 9:
                for _k in range(2):
                    s0 = int(rng.integers(0, n - 8))
                    p[s0:s0 + 9] = kmers[int(rng.integers(0, 40))]   # few distinct kmers => repeats survive -l 2
            if n and rng.random() < 0.1:
                p[int(rng.integers(0, n))] = ord("*")
            frags.append(bytes(p))
        data, offs = api._concat(frags)
        n_reads = 400
        hits, roffs = [], [0]
        for r in range(n_reads):
            for f in frags[r * F:(r + 1) * F]:
                hits += [d.get(f[i:i + 9], 0) for i in range(max(0, len(f) - 8))]
            roffs.append(len(hits))
        d_data, d_offs = torch.from_numpy(data.copy()).cuda(), torch.from_numpy(offs.view(np.int64)).cuda()
        for method, strategy, lb in (("tree", "lca*", 2.0), ("rmq", "mrtl", 1.0), ("tree", "hybrid", 2.0), ("rmq", "hybrid", 0.0)):
            d_t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
            d_s = torch.full_like(d_t, -1)
            api.classify_prot9_dev(ix, tx, d_data, d_offs, n_reads, F, d_t, d_s, method=method, strategy=strategy, factor=0.4, lower_bound=lb)
            torch.cuda.synchronize()
            t, s = d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32)
            agg = O.Aggregator(otx, method, strategy, 0.4)
            nbad, bad, wt, ws = agg.taxa2agg_check(np.array(hits, np.uint32), np.array(roffs, np.uint64), t, s, lower_bound=lb)
            assert nbad == 0, (F, method, strategy, bad)


def test_classify_prot9_overlong_reads_are_flagged(api, torch_cuda):
    """A read of more than 128 windows or 384 bytes is UMGAP_READ_UNSUPPORTED (include/umgap_hip.h) with every strategy, and
    the reads that share its group are unaffected."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    kmers = sorted({rng.choice(letters, 9).tobytes() for _ in range(500)})
    ids = np.arange(1, 60, dtype=np.uint32)
    par = np.maximum(1, ids // 3).astype(np.uint32)
    tx = api.Taxonomy(ids, par)
    otx = O.Taxonomy.from_arrays(ids, par)
    vals = rng.choice(ids, len(kmers)).astype(np.uint32)
    keys, _ = api.pack9_array(kmers)
    ix = api.Index.build(keys, vals)
    d = dict(zip(kmers, vals.tolist()))
    F, n_reads = 2, 203
    frags, long_read = [], np.zeros(n_reads, bool)
    for r in range(n_reads):
        kind = r % 7
        lens = (70, 70) if kind == 3 else ((300, 100) if kind == 5 else (int(rng.integers(9, 60)), int(rng.integers(0, 60))))   # 124 windows; 400 bytes
        if r % 29 == 0:
            lens = (137, 0)                                                                                                    # 129 windows
        long_read[r] = sum(max(0, n - 8) for n in lens) > 128 or sum(lens) > 384
        for n in lens:
            p = bytearray(rng.choice(letters, n).tobytes())
            for _k in range(n // 12):
                s0 = int(rng.integers(0, n - 8))
                p[s0:s0 + 9] = kmers[int(rng.integers(0, 25))]
            frags.append(bytes(p))
    assert long_read.sum() > 20 and (~long_read).sum() > 100
    data, offs = api._concat(frags)
    d_data, d_offs = torch.from_numpy(data.copy()).cuda(), torch.from_numpy(offs.view(np.int64)).cuda()
    hits, roffs = [], [0]
    for r in range(n_reads):
        if not long_read[r]:
            for f in frags[r * F:(r + 1) * F]:
                hits += [d.get(f[i:i + 9], 0) for i in range(max(0, len(f) - 8))]
        roffs.append(len(hits))
    for method, strategy in (("tree", "lca*"), ("rmq", "mrtl"), ("tree", "hybrid")):
        d_t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
        d_s = torch.full_like(d_t, -1)
        api.classify_prot9_dev(ix, tx, d_data, d_offs, n_reads, F, d_t, d_s, method=method, strategy=strategy, factor=0.25, lower_bound=1.0)
        torch.cuda.synchronize()
        t, s = d_t.cpu().numpy().view(np.uint32).copy(), d_s.cpu().numpy().view(np.uint32).copy()
        assert (s[long_read] == 4).all() and (t[long_read] == 0).all(), (method, strategy)
        assert (s[~long_read] != 4).all()
        agg = O.Aggregator(otx, method, strategy, 0.25)
        _, _, wt, ws = agg.taxa2agg_check(np.array(hits, np.uint32), np.array(roffs, np.uint64), lower_bound=1.0)
        t[long_read], s[long_read] = wt[long_read], ws[long_read]           # the flagged reads are empty records on the oracle side
        nbad, bad, _, _ = agg.taxa2agg_check(np.array(hits, np.uint32), np.array(roffs, np.uint64), t, s, lower_bound=1.0)
        assert nbad == 0, (method, strategy, bad)


def _tryptic_pieces(seq):
    """The reference's digest of one sequence (src/commands/prot2tryp2lca.rs:107-116), literally: the regex applied twice, `*`
    to newline, empty lines dropped."""
    import re
    pat = re.compile(rb"([KR])([^P])")
    once = pat.sub(rb"\1\n\2", seq)
    twice = pat.sub(rb"\1\n\2", once)
    return [p for p in twice.replace(b"*", b"\n").split(b"\n") if p]


def test_digest_tryp_dev_matches_the_regex(api, torch_cuda):
    """umgap_digest_tryp_dev against the reference's regex digest: K/R runs, KP / RP, `*` (single, runs, at the ends), empty and
    one-byte fragments, fragments that end in K, a P that opens a fragment -- and sizes that straddle the 2048-byte tiles."""
    torch = torch_cuda
    rng = np.random.default_rng(31)
    alphabet = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYKRKRPP**", np.uint8)
    special = [b"", b"K", b"P", b"*", b"**", b"KP", b"KK", b"KRP", b"KKP", b"RPK", b"AK", b"KA", b"*K*", b"PKPKP", b"AAAK*", b"*AAAK", b"KR" * 40,
               b"A" * 5000, b"K" * 3000, b"AKP" * 900]
    for trial, n_frag in enumerate((1, 7, 400, 3000)):
        frags = [special[i % len(special)] if i % 3 == 0 else rng.choice(alphabet, int(rng.integers(0, 120))).tobytes() for i in range(n_frag)]
        data, offs = api._concat(frags)
        pad = rng.integers(0, 9)                      # the byte range need not start at 0
        d_bytes = torch.from_numpy(np.concatenate([np.full(pad, ord("K"), np.uint8), data, np.full(3, ord("K"), np.uint8)])).cuda()
        d_offs = torch.from_numpy((offs + np.uint64(pad)).view(np.int64)).cuda()
        got = api.digest_tryp_dev(d_bytes, d_offs).cpu().numpy().view(np.uint64)
        host = d_bytes.cpu().numpy().tobytes()
        pieces = [host[int(a):int(b)] for a, b in zip(got[:-1], got[1:])]
        whole = b"".join(frags)
        assert b"".join(pieces) == whole and int(got[0]) == pad and int(got[-1]) == pad + len(whole), trial
        assert all(len(p) > 0 for p in pieces)
        want = [p for f in frags for p in _tryptic_pieces(f)]
        assert [p for p in pieces if p != b"*"] == want, trial
        assert all(p == b"*" or b"*" not in p for p in pieces)
    # offsets that go backwards are refused, not followed
    bad = torch.tensor([0, 10, 5, 20], dtype=torch.int64, device="cuda")
    with pytest.raises(api.UmgapError, match="must not decrease"):
        api.digest_tryp_dev(torch.zeros(32, dtype=torch.uint8, device="cuda"), bad)
    assert api.digest_tryp_dev(torch.zeros(4, dtype=torch.uint8, device="cuda"), torch.tensor([2, 2, 2], dtype=torch.int64, device="cuda")).tolist() == [2]


@pytest.mark.parametrize("mode", [("tree", "hybrid", 0.25, 0.0), ("rmq", "mrtl", 0.0, 1.0), ("tree", "lca*", 0.0, 2.0), ("rmq", "hybrid", 0.4, 1.0)])
@pytest.mark.parametrize("chunks", [0, 3])
def test_classify_tryp_dev_vs_oracle(api, torch_cuda, mode, chunks, monkeypatch):
    """umgap_classify_tryp_dev = digest + length window + lookup + aggregate, against the regex digest, a dict and the oracle's
    taxa2agg -- 1..3 fragments per read, windows 5..50 and 1..63.  chunks: the call cut into chunks of whole reads over two lanes
    (UMGAP_TRYP_CHUNKS, an opt-in experiment: DESIGN.md section 4)."""
    torch = torch_cuda
    if chunks:
        monkeypatch.setenv("UMGAP_TRYP_CHUNKS", str(chunks))
    method, strategy, factor, lb = mode
    rng = np.random.default_rng(17)
    letters = np.frombuffer(b"ACDEFGHILMNPQSTVWY", np.uint8)

    def peptide():
        k = int(rng.integers(1, 30))
        p = bytearray(rng.choice(letters, k).tobytes())
        p[-1] = ord("K") if rng.random() < 0.5 else ord("R")
        if p[0] == ord("P"):
            p[0] = ord("A")
        return bytes(p)
    peps = sorted({peptide() for _ in range(6000)} | {b"*"})           # a `*` key must never be found through the digest
    ids = np.arange(1, 200, dtype=np.uint32)
    par = np.maximum(1, ids // 2).astype(np.uint32)
    tx = api.Taxonomy(ids, par)
    otx = O.Taxonomy.from_arrays(ids, par)
    vals = rng.choice(ids, len(peps)).astype(np.uint32)
    ix = api.Index.build_var(peps, vals)
    d = dict(zip(peps, vals.tolist()))
    pool = [p for p in peps if p != b"*"][:60]                           # few distinct peptides per read survive -l
    for F, (lo, hi) in ((1, (5, 50)), (3, (1, 63)), (2, (9, 45))):
        n_reads = 700
        frags = []
        for _ in range(n_reads * F):
            parts = []
            for _k in range(int(rng.integers(0, 7))):
                u = rng.random()
                parts.append(pool[int(rng.integers(0, len(pool)))] if u < 0.6 else (peptide() if u < 0.85 else (b"*" if u < 0.93 else b"P" + peptide())))
            frags.append(b"".join(parts))
        data, offs = api._concat(frags)
        d_bytes, d_offs = torch.from_numpy(np.concatenate([data, np.zeros(1, np.uint8)])).cuda(), torch.from_numpy(offs.view(np.int64)).cuda()
        d_t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
        d_s = torch.full_like(d_t, -1)
        n_pieces = api.classify_tryp_dev(ix, tx, d_bytes, d_offs, n_reads, F, d_t, d_s, method=method, strategy=strategy, factor=factor, lower_bound=lb,
                                         min_len=lo, max_len=hi)
        hits, roffs = [], [0]
        for r in range(n_reads):
            for f in frags[r * F:(r + 1) * F]:
                hits += [d.get(p, 0) for p in _tryptic_pieces(f) if lo <= len(p) <= hi]
            roffs.append(len(hits))
        assert n_pieces >= len(hits)
        t, s = d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32)
        agg = O.Aggregator(otx, method, strategy, factor)
        nbad, bad, wt, ws = agg.taxa2agg_check(np.array(hits, np.uint32), np.array(roffs, np.uint64), t, s, lower_bound=lb)
        assert nbad == 0, (F, lo, hi, bad)
        assert (s == 0).sum() > 10 and (s == 0).mean() > (0.3 if lb < 2 else 0.0)
    nine = api.Index.build(np.array([5], np.uint64), np.array([7], np.uint32))
    with pytest.raises(api.UmgapError, match="variable-length"):
        api.classify_tryp_dev(nine, tx, d_bytes, d_offs, 1, 1, d_t, d_s)
    with pytest.raises(api.UmgapError, match="packed peptide table"):
        ix.bind(tx)                                                      # the `*` key keeps this table in the heap layout


@pytest.mark.parametrize("mode", [("tree", "hybrid", 0.25, 0.0), ("rmq", "mrtl", 0.0, 1.0), ("tree", "lca*", 0.0, 2.0), ("rmq", "hybrid", 0.4, 1.0), ("rmq", "mrtl", 0.0, 5.0)])
def test_classify_tryp_dev_on_a_bound_peptide_table(api, torch_cuda, mode):
    """umgap_index_bind_taxonomy on the packed peptide table: the header slots' taxa become preorder-interval codes and
    umgap_classify_tryp_dev aggregates them with no by-id gather.  Same answers, read for read, as on the unbound table (and so as
    the oracle, test above) -- short reads, reads of hundreds of peptides (the large-scratch passes, which get ids back), every
    strategy; the plain look-ups refuse a bound table; unbinding gives the table back byte for byte."""
    torch = torch_cuda
    method, strategy, factor, lb = mode
    rng = np.random.default_rng(29)
    letters = np.frombuffer(b"ACDEFGHILMNPQSTVWY", np.uint8)

    def peptide():
        k = int(rng.integers(2, 30))
        p = bytearray(rng.choice(letters, k).tobytes())
        p[-1] = ord("K") if rng.random() < 0.5 else ord("R")
        if p[0] == ord("P"):
            p[0] = ord("A")
        return bytes(p)
    peps = sorted({peptide() for _ in range(6000)})
    ids = np.arange(1, 3000, dtype=np.uint32)
    par = np.maximum(1, ids // 2).astype(np.uint32)
    tx = api.Taxonomy(ids, par)
    vals = rng.choice(ids, len(peps)).astype(np.uint32)
    ix = api.Index.build_var(peps, vals)
    assert ix.var_layout == 1
    before = ix.dump_var()
    pool, wide = peps[:60], peps[:1500]
    for F, (lo, hi), long_reads in ((2, (5, 50), False), (1, (1, 63), True), (3, (9, 45), False)):
        n_reads = 600
        frags = []
        for i in range(n_reads * F):
            n_parts = int(rng.integers(0, 7))
            src = pool
            if long_reads and i % 50 == 0:
                n_parts, src = int(rng.integers(100, 400)), wide          # > 64 distinct taxa and / or > 128 hits: the deferred passes
            parts = []
            for _k in range(n_parts):
                u = rng.random()
                parts.append(src[int(rng.integers(0, len(src)))] if u < 0.7 else (peptide() if u < 0.9 else b"*"))
            frags.append(b"".join(parts))
        data, offs = api._concat(frags)
        d_bytes, d_offs = torch.from_numpy(np.concatenate([data, np.zeros(1, np.uint8)])).cuda(), torch.from_numpy(offs.view(np.int64)).cuda()
        got = []
        for bound in (False, True):
            if bound:
                ix.bind(tx)
            d_t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
            d_s = torch.full_like(d_t, -1)
            n_pieces = api.classify_tryp_dev(ix, tx, d_bytes, d_offs, n_reads, F, d_t, d_s, method=method, strategy=strategy, factor=factor, lower_bound=lb,
                                             min_len=lo, max_len=hi)
            got.append((n_pieces, d_t.cpu().numpy().copy(), d_s.cpu().numpy().copy()))
            if bound:
                with pytest.raises(api.UmgapError, match="unbind it first"):
                    ix.lookup_var([peps[0]])
                ix.unbind(tx)
        assert got[0][0] == got[1][0]
        assert np.array_equal(got[0][2], got[1][2]), (F, "status")
        assert np.array_equal(got[0][1], got[1][1]), (F, "taxon")
        assert lb >= 2 or (got[0][2] == 0).mean() > 0.3
    after = ix.dump_var()
    assert all(np.array_equal(a, b) for a, b in zip(before, after))
    other = api.Taxonomy(ids[:100], par[:100])
    with pytest.raises(api.UmgapError, match="not taxa of this taxonomy"):
        ix.bind(other)


def test_classify_records_long_ragged_reads(api, torch_cuda):
    """umgap_classify_records (host records in, one taxon per read out; what `umgap prot2kmer2agg` calls per block) on reads
    the fused kernels do not take: 1..12 fragments of 9..700 residues (6-frame translations, whole proteins), every
    strategy, with and without seedextend -- against the oracle stages."""
    rng = np.random.default_rng(123)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    from umgap_amd.synth import synthetic_taxonomy
    ids, par, valid, rank = synthetic_taxonomy(3000, 5000, max_depth=14, seed=21)
    tx = api.Taxonomy(ids, par, valid, rank)
    otx = O.Taxonomy.from_arrays(ids, par, valid, rank)
    parent_of = dict(zip(ids.tolist(), par.tolist()))
    proteins, table = [], {}
    for _ in range(120):
        prot = rng.choice(letters, int(rng.integers(60, 900))).tobytes()
        t = int(rng.choice(ids))
        proteins.append(prot)
        for i in range(len(prot) - 8):
            table.setdefault(prot[i:i + 9], parent_of[t] if (i // 11) % 4 == 3 else t)
    kmers = sorted(table)
    keys, _ = api.pack9_array(kmers)
    ix = api.Index.build(keys, np.array([table[k] for k in kmers], np.uint32))
    frags, read_frags = [], [0]
    for r in range(400):
        for _ in range(int(rng.integers(1, 13))):
            prot = proteins[int(rng.integers(0, len(proteins)))]
            a = int(rng.integers(0, max(1, len(prot) - 9)))
            f = bytearray(prot[a:a + int(rng.integers(9, 700))])
            for i in range(len(f)):
                m = rng.random()
                if m < 0.02:
                    f[i] = int(rng.choice(letters))
                elif m < 0.03:
                    f[i] = ord("*")
            if len(f) >= 9:
                frags.append(bytes(f))
        read_frags.append(len(frags))
    assert max(len(f) for f in frags) > 300
    for se in (False, True):
        hits, roffs = [], [0]
        for r in range(len(read_frags) - 1):
            for f in frags[read_frags[r]:read_frags[r + 1]]:
                h = np.array([table.get(f[i:i + 9], 0) for i in range(len(f) - 8)], np.uint32)
                if se:
                    kept = O.seedextend_record(h, 3, 1)
                    h = kept
                hits += h.tolist()
            roffs.append(len(hits))
        hits, roffs = np.array(hits, np.uint32), np.array(roffs, np.uint64)
        for method, strategy, lb in (("tree", "lca*", 2.0), ("tree", "hybrid", 1.0), ("rmq", "mrtl", 1.0), ("rmq", "hybrid", 0.0), ("rmq", "lca*", 5.0)):
            t, s = api.classify_records(ix, tx, frags, read_frags, method=method, strategy=strategy, factor=0.25, lower_bound=lb,
                                        seedextend=se, min_seed_size=3, max_gap_size=1)
            agg = O.Aggregator(otx, method, strategy, factor=0.25)
            nbad, bad, wt, ws = agg.taxa2agg_check(hits, roffs, t, s, lower_bound=lb, nthreads=8)
            assert nbad == 0, (se, method, strategy, bad, t[bad.astype(int)], wt[bad.astype(int)])
            assert (s == 0).mean() > 0.5


@pytest.mark.parametrize("F", [2, 0])
def test_classify_records_residues_in_pieces(api, torch_cuda, F):
    """umgap_classify_records_sg: the residues handed over as pieces laid end to end (what a host that tokenizes on several threads
    holds) -- pageable pieces are gathered into the staging, page-locked ones are copied from where they lie; fragments may run
    across pieces, pieces may be empty, a chunk of the call may start and end anywhere in a piece.  Same answers as the one-array
    call, bit for bit (fused route F = 2; general route: ragged reads)."""
    import torch
    rng = np.random.default_rng(77 + F)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    from umgap_amd.synth import synthetic_taxonomy
    ids, par, valid, rank = synthetic_taxonomy(3000, 5000, max_depth=14, seed=23)
    tx = api.Taxonomy(ids, par, valid, rank)
    prot = rng.choice(letters, 60_000).astype(np.uint8)
    windows = np.lib.stride_tricks.sliding_window_view(prot, 9)
    block_taxon = rng.choice(ids, len(prot) // 300 + 1)      # stretches of 300 residues under one taxon each (seedextend wants runs)
    table = {}
    for i, w in enumerate(windows):
        table.setdefault(w.tobytes(), int(block_taxon[i // 300]))
    kmers = sorted(table)
    keys, _ = api.pack9_array(kmers)
    ix = api.Index.build(keys, np.array([table[k] for k in kmers], np.uint32))
    n_reads = 30_000
    nfr = np.full(n_reads, F, np.int64) if F else rng.integers(0, 4, n_reads)
    read_frags = np.concatenate([[0], np.cumsum(nfr)]).astype(np.uint64)
    n_frag = int(read_frags[-1])
    flen = rng.choice([9, 20, 33, 50], n_frag)
    start = rng.integers(0, len(prot) - 50, n_frag)
    offs = np.concatenate([[0], np.cumsum(flen)]).astype(np.uint64)
    data = np.empty(int(offs[-1]), np.uint8)
    for f in range(n_frag):
        data[int(offs[f]):int(offs[f + 1])] = prot[start[f]:start[f] + flen[f]]
    noise = rng.random(len(data)) < 0.02
    data[noise] = rng.choice(letters, int(noise.sum()))
    kw = dict(method="tree", strategy="lca*", lower_bound=2.0, seedextend=True, min_seed_size=2, max_gap_size=1)
    want_t, want_s = api.classify_records_arrays(ix, tx, data, offs, read_frags, **kw)
    assert (want_s == 0).mean() > 0.3
    os.environ["UMGAP_RECORDS_CHUNK_MB"] = "1"      # ~3 MB of residues: chunks that begin and end inside pieces
    try:
        for how in ("one", "few", "many with empty ones", "single bytes at the seams"):
            if how == "one":
                cuts = []
            elif how == "few":
                cuts = sorted(rng.integers(0, len(data), 5).tolist())
            elif how == "many with empty ones":
                cuts = sorted(rng.integers(0, len(data), 40).tolist() + [0, 0, len(data), 1000, 1000])
            else:
                c = int(offs[n_frag // 2])
                cuts = [c - 1, c, c + 1, c + 2]
            edges = [0] + cuts + [len(data)]
            for pinned in (False, True):
                pieces = []
                for a, b in zip(edges[:-1], edges[1:]):
                    if pinned and b > a:
                        t = torch.empty(b - a, dtype=torch.uint8).pin_memory()
                        t.numpy()[:] = data[a:b]
                        pieces.append(t)
                    else:
                        pieces.append(data[a:b].copy())
                t, s = api.classify_records_pieces(ix, tx, [x.numpy() if isinstance(x, torch.Tensor) else x for x in pieces], offs, read_frags, **kw)
                assert (t == want_t).all() and (s == want_s).all(), (how, pinned, np.flatnonzero((t != want_t) | (s != want_s))[:5])
    finally:
        del os.environ["UMGAP_RECORDS_CHUNK_MB"]
    # pieces that end before the last fragment does / ends that decrease: refused
    with pytest.raises(api.UmgapError, match="pieces end before"):
        api.classify_records_pieces(ix, tx, [data[:-5]], offs, read_frags, **kw)
    import ctypes
    from umgap_amd import _lib
    ptrs = (ctypes.c_void_p * 2)(data.ctypes.data, data.ctypes.data)
    ends = np.array([len(data), 10], np.uint64)
    out = np.zeros(n_reads, np.uint32)
    rc = _lib.load().umgap_classify_records_sg(ix._h, tx._h, 0, 0, 0.25, 0.0, 0, ctypes.cast(ptrs, ctypes.c_void_p), ends.ctypes.data_as(ctypes.c_void_p), 2,
                                               offs.ctypes.data_as(ctypes.c_void_p), n_frag, read_frags.ctypes.data_as(ctypes.c_void_p), n_reads, 1, 2, 0,
                                               out.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == _lib.ERR_INVALID


@pytest.mark.parametrize("F,misfit", [(2, None), (1, None), (6, None), (2, "long read"), (2, "ragged"), (3, "many taxa"), (12, "six frames"), (12, "six frames, all coding"),
                                      (10, "six frames")])
def test_classify_records_fused_route(api, torch_cuda, F, misfit):
    """umgap_classify_records takes a call whose reads all have F fragments through ONE fused launch per chunk (k_classify_prot9;
    records.hip: k_records_fit) and falls back to its general kernels for the whole call when a read does not fit the fused kernel
    (too long, another number of fragments, more distinct taxa than its scratch holds): either way the answers must be those of
    the general kernels (UMGAP_RECORDS_FUSED=0) and of the oracle stages -- every strategy, seedextend on and off, min_seed_size 1
    (where the reference's seedextend can panic) included."""
    rng = np.random.default_rng(500 + F + (len(misfit) if misfit else 0))
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    from umgap_amd import _lib
    from umgap_amd.synth import synthetic_taxonomy
    ids, par, valid, rank = synthetic_taxonomy(3000, 5000, max_depth=14, seed=22)
    tx = api.Taxonomy(ids, par, valid, rank)
    otx = O.Taxonomy.from_arrays(ids, par, valid, rank)
    proteins, table = [], {}
    for _ in range(150):
        prot = rng.choice(letters, int(rng.integers(40, 200)) if misfit != "many taxa" else 400).tobytes()
        t = int(rng.choice(ids))
        proteins.append(prot)
        for i in range(len(prot) - 8):
            table.setdefault(prot[i:i + 9], int(rng.choice(ids)) if misfit == "many taxa" else t)
    kmers = sorted(table)
    keys, _ = api.pack9_array(kmers)
    ix = api.Index.build(keys, np.array([table[k] for k in kmers], np.uint32))
    n_reads = 900
    frags, read_frags = [], [0]
    for r in range(n_reads):
        nf = F
        if misfit == "ragged" and r == 611:
            nf = F + 1
        for k in range(nf):
            prot = proteins[int(rng.integers(0, len(proteins)))]
            a = int(rng.integers(0, max(1, len(prot) - 9)))
            n = int(rng.choice([0, 5, 9, 20, 33, 50] if F < 4 else [0, 5, 9, 20, 28]))      # (at most 128 windows per read)
            if misfit and misfit.startswith("six frames"):
                # a pair translated in all six frames (translate -a, scripts/umgap-analyse.sh:277-288): fragments of ~50 residues, ~500
                # windows per read -- the LONG variant of the fused kernel; the hits sit in the frames that code (here: two of them, the
                # other fragments are random residues), unless "all coding"
                n = int(rng.choice([50, 50, 50, 41, 9, 0]))
                if misfit == "six frames" and k % (F // 2) != 0:
                    frags.append(rng.choice(letters, n).tobytes())
                    continue
            if misfit == "long read" and r == 77 and k == 0:      # longer than the LONG variant of the fused kernel takes (1536 residues)
                n, a = 1700, 0
                prot = prot * 50
            if misfit == "many taxa":
                n = 50                                # 126 windows per read, each 9-mer under a taxon of its own
            f = bytearray(prot[a:a + n])
            for i in range(len(f)):
                if rng.random() < 0.03:
                    f[i] = int(rng.choice([int(rng.choice(letters)), 42]))
            frags.append(bytes(f))
        read_frags.append(len(frags))
    if misfit == "long read":
        assert sum(len(f) for f in frags[read_frags[77]:read_frags[78]]) > 1536
    for se, ms in ((False, 2), (True, 3), (True, 1)):
        hits, roffs, panics = [], [0], set()
        for r in range(n_reads):
            for f in frags[read_frags[r]:read_frags[r + 1]]:
                h = np.array([table.get(f[i:i + 9], 0) for i in range(max(0, len(f) - 8))], np.uint32)
                if se and len(h):
                    h = O.seedextend_record(h, ms, 1)
                    if h is None:           # the reference's slice panic (src/commands/seedextend.rs:169-173)
                        panics.add(r)
                        h = np.zeros(0, np.uint32)
                hits += h.tolist()
            roffs.append(len(hits))
        hits, roffs = np.array(hits, np.uint32), np.array(roffs, np.uint64)
        for method, strategy, lb in (("tree", "lca*", 2.0), ("tree", "hybrid", 0.0), ("rmq", "mrtl", 1.0)):
            kw = dict(method=method, strategy=strategy, factor=0.25, lower_bound=lb, seedextend=se, min_seed_size=ms, max_gap_size=1)
            os.environ["UMGAP_RECORDS_FUSED"] = "0"
            try:
                t0, s0 = api.classify_records(ix, tx, frags, read_frags, **kw)
            finally:
                del os.environ["UMGAP_RECORDS_FUSED"]
            assert _lib.load().umgap_records_last_route() == 0
            t1, s1 = api.classify_records(ix, tx, frags, read_frags, **kw)
            route = _lib.load().umgap_records_last_route()
            if misfit is None or misfit == "six frames":
                assert route == 1, (misfit, strategy, se, route, np.bincount(s1, minlength=5))
            elif misfit == "six frames, all coding":
                assert route == 2 or se     # ~400 hits left in a read where the aggregate stage takes 128 (seedextend may leave fewer)
            elif misfit == "ragged":
                assert route == 0              # the totals already say so
            elif misfit == "long read":
                assert route == 2, (misfit, strategy, route, np.bincount(s1, minlength=5))
            elif misfit == "many taxa":
                # ~95 distinct taxa in a read where the group aggregate of the fused kernel holds 64: its large-scratch pass (128) takes
                # the read since round 4, so the call stays on the fused route
                assert route == 1, (misfit, strategy, route, np.bincount(s1, minlength=5))
            same = (s0 == s1) & ((t0 == t1) | (s0 != 0))
            if strategy == "hybrid":          # ties between children may be broken either way (tie sets: the oracle check below)
                same |= (s0 == s1)
            assert same.all(), (se, ms, method, strategy, np.flatnonzero(~same)[:5])
            agg = O.Aggregator(otx, method, strategy, factor=0.25)
            nbad, bad, wt, ws = agg.taxa2agg_check(hits, roffs, t1, s1, lower_bound=lb, nthreads=8)
            bad = np.array([b for b in np.asarray(bad).astype(int).tolist() if b not in panics], int)
            assert len(bad) == 0, (se, ms, method, strategy, bad[:5], t1[bad[:5]], wt[bad[:5]])
            for r in panics:
                assert s1[r] == s0[r] == 3, (r, s1[r])      # UMGAP_READ_PANIC
            # the table bound to the taxonomy (how `umgap prot2kmer2agg` runs it): both routes, bit for bit the unbound answers
            ix.bind(tx)
            try:
                os.environ["UMGAP_RECORDS_FUSED"] = "0"
                try:
                    tb0, sb0 = api.classify_records(ix, tx, frags, read_frags, **kw)
                finally:
                    del os.environ["UMGAP_RECORDS_FUSED"]
                tb1, sb1 = api.classify_records(ix, tx, frags, read_frags, **kw)
                assert _lib.load().umgap_records_last_route() == route
            finally:
                ix.unbind(tx)
            assert (tb0 == t0).all() and (sb0 == s0).all(), (se, ms, method, strategy, "general route, bound")
            assert (tb1 == t1).all() and (sb1 == s1).all(), (se, ms, method, strategy, "fused route, bound")


@pytest.mark.parametrize("layout", ["plain", "walk order"])
def test_lookup_records_cuts_the_windows_on_the_device(api, torch_cuda, layout):
    """umgap_index_lookup_records (what `umgap prot2kmer2lca` runs on): residues in host memory in, the hit of every window out,
    indexed by the byte position the window starts at.  Against a dict on fragments of every length (empty, shorter than 9, odd
    bytes inside), more short fragments than a chunk's offset table holds, and one fragment longer than a chunk (the chunks then
    cut inside it and bring 8 bytes of lookahead along), from pageable and from pinned arrays."""
    torch = torch_cuda
    rng = np.random.default_rng(31)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    n_tax = 5000
    prot = rng.choice(letters, 600_000).astype(np.uint8)
    table = {}
    pk, pok = _pack_windows(prot.reshape(1, -1))
    vals = rng.integers(1, n_tax, pk.shape[1]).astype(np.uint32)
    for k, v in zip(pk[0].tolist(), vals.tolist()):
        table.setdefault(k, v)
    keys = np.array(sorted(table), np.uint64)
    kv = np.array([table[int(k)] for k in keys], np.uint32)
    if layout == "plain":
        ix = api.Index.build(keys, kv)
    else:
        values = api.Index.build(keys, kv)
        d_prot = torch.from_numpy(prot).cuda()
        d_off = torch.tensor([0, len(prot)], dtype=torch.int64, device="cuda")
        ix = api.Index.build_walk(values, [(d_prot, d_off, len(prot))])
        assert ix.kind == 10
    # fragments: slices of the proteome (hits) and random residues (misses), every length; then 300 k tiny ones; then one of 9 MB
    frags = []
    for _ in range(3000):
        n = int(rng.choice([0, 1, 8, 9, 10, 17, 50, 50, 200]))
        a = int(rng.integers(0, len(prot) - 200))
        f = prot[a:a + n].copy() if rng.random() < 0.7 else rng.choice(letters, n).astype(np.uint8)
        if n > 10 and rng.random() < 0.1:
            f[int(rng.integers(0, n))] = int(rng.choice([42, 120, 0xC3]))
        frags.append(f)
    frags += [prot[a:a + int(n)] for a, n in zip(rng.integers(0, len(prot) - 12, 300_000), rng.integers(0, 12, 300_000))]
    frags.append(np.tile(prot, 15))
    frags += [prot[5:60], np.zeros(0, np.uint8), prot[100:109]]
    lens = np.array([len(f) for f in frags], np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64) + np.uint64(7)          # (the first fragment need not start at 0)
    data = np.concatenate([np.zeros(7, np.uint8)] + frags)
    got = ix.lookup_records(data, offs, miss_value=0)
    # expected: per fragment, per window
    want = np.zeros(len(data) - 7, np.uint32)
    for f, o in zip(frags, (offs[:-1] - np.uint64(7)).tolist()):
        if len(f) < 9:
            continue
        k, ok = _pack_windows(f.reshape(1, -1))
        pos = np.searchsorted(keys, k[0])
        pos[pos >= len(keys)] = 0
        want[o:o + len(f) - 8] = np.where(ok[0] & (keys[pos] == k[0]), kv[pos], 0)
    assert (got == want).all(), np.flatnonzero(got != want)[:10]
    assert (want != 0).mean() > 0.3
    # pinned arrays in and out: copied from / to directly
    h_data = torch.from_numpy(data).pin_memory()
    h_offs = torch.from_numpy(offs.view(np.int64)).pin_memory()
    h_out = torch.empty(len(want), dtype=torch.int32).pin_memory()
    from umgap_amd import _lib
    _lib.check(_lib.load().umgap_index_lookup_records(ix._h, h_data.data_ptr(), h_offs.data_ptr(), len(frags), 5, h_out.data_ptr()))
    assert (h_out.numpy().view(np.uint32) == np.where(want == 0, 5, want)).all()


def test_host_pointer_calls_across_chunk_seams(api, torch_cuda):
    """The host-pointer entry points cut a call into chunks that travel over an upload lane while earlier chunks compute
    (records.hip, index.hip, index_var.hip, aggregate.hip): the answers must not depend on where the seams fall.  Each call is
    made on inputs that span several chunks -- from pageable and from pinned arrays -- and compared with the device-pointer call
    on the same data."""
    torch = torch_cuda
    from umgap_amd import _lib
    from umgap_amd.api import METHODS, STRATEGIES, Index, check, classify_records
    from umgap_amd.synth import SyntheticWorkload
    L = _lib.load()
    rng = np.random.default_rng(77)
    w = SyntheticWorkload(n_tax=20_000, n_keys=3_000_000, build_index=True)
    # -- get(): 17.6 M keys = five chunks of 4 M
    d_q = w.reads(0, 210_000)
    want = w.index.lookup9_dev(d_q, miss_value=5).cpu()
    for pin in (False, True):
        h_q = torch.empty(d_q.numel(), dtype=torch.int64, pin_memory=pin)
        h_q.copy_(d_q.cpu())
        h_o = torch.full((d_q.numel(),), -1, dtype=torch.int32, pin_memory=pin)
        check(L.umgap_index_lookup9(w.index._h, h_q.data_ptr(), d_q.numel(), 5, h_o.data_ptr()))
        assert torch.equal(h_o, want), pin
    # -- aggregate: 17 M taxa in ragged reads = three chunks of 8 M taxa
    d_t = w.index.lookup9_dev(d_q, miss_value=0)
    lens = rng.integers(0, 170, 300_000)
    offs = np.zeros(len(lens) + 1, np.int64)
    offs[1:] = np.cumsum(lens)
    n_reads = int(np.searchsorted(offs, d_t.numel(), side="right") - 1)
    offs = offs[:n_reads + 1]
    assert offs[-1] > 16_800_000
    d_o = torch.from_numpy(offs).cuda()
    for method, strategy, lb in (("tree", "lca*", 2.0), ("rmq", "mrtl", 1.0), ("tree", "hybrid", 0.0)):
        t0 = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
        s0 = torch.full_like(t0, -1)
        w.taxonomy.aggregate_dev(d_t, d_o, t0, s0, method=method, strategy=strategy, factor=0.25, lower_bound=lb)
        torch.cuda.synchronize()
        for pin in (False, True):
            h_t = torch.empty(d_t.numel(), dtype=torch.int32, pin_memory=pin)
            h_t.copy_(d_t.cpu())
            h_off = torch.from_numpy(offs.copy())
            o_t, o_s = torch.full((n_reads,), -2, dtype=torch.int32), torch.full((n_reads,), -2, dtype=torch.int32)
            if pin:
                h_off, o_t, o_s = h_off.pin_memory(), o_t.pin_memory(), o_s.pin_memory()
            check(L.umgap_aggregate(w.taxonomy._h, METHODS[method], STRATEGIES[strategy], 0.25, lb, 0, h_t.data_ptr(), None, h_off.data_ptr(), n_reads, o_t.data_ptr(), o_s.data_ptr()))
            assert torch.equal(o_s, s0.cpu()), (strategy, pin)
            same = (o_t == t0.cpu()) | (o_s != 0)
            assert same.all() if strategy != "hybrid" else same.float().mean() > 0.99, (strategy, pin)     # (hybrid: ties between children)
    # -- records: 6 MB of residues in chunks of 1 MB, both routes, against one chunk
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    prot = [rng.choice(letters, 60).tobytes() for _ in range(3000)]
    table = {}
    for i, p in enumerate(prot):
        for k in range(len(p) - 8):
            table.setdefault(p[k:k + 9], int(w.ids[i % len(w.ids)]))
    kmers = sorted(table)
    keys, _ = api.pack9_array(kmers)
    ix = Index.build(keys, np.array([table[k] for k in kmers], np.uint32))
    frags = []
    for r in range(60_000):
        for _ in range(2):
            f = bytearray(prot[int(rng.integers(0, len(prot)))][int(rng.integers(0, 8)):][:50])
            if rng.random() < 0.3:
                f[int(rng.integers(0, len(f)))] = int(rng.choice(letters))
            frags.append(bytes(f))
    rf = np.arange(0, len(frags) + 1, 2, dtype=np.uint64)
    kw = dict(method="tree", strategy="lca*", lower_bound=2.0, seedextend=True, min_seed_size=3, max_gap_size=1)
    ref = classify_records(ix, w.taxonomy, frags, rf, **kw)
    assert L.umgap_records_last_route() == 1 and (ref[1] == 0).mean() > 0.5
    for fused in ("1", "0"):
        os.environ["UMGAP_RECORDS_CHUNK_MB"], os.environ["UMGAP_RECORDS_FUSED"] = "1", fused
        try:
            got = classify_records(ix, w.taxonomy, frags, rf, **kw)
        finally:
            del os.environ["UMGAP_RECORDS_CHUNK_MB"], os.environ["UMGAP_RECORDS_FUSED"]
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), fused
    # -- tryptic get(): 2.5 M peptides = three chunks of 1 M
    pep = [prot[int(i)][int(a):int(a) + int(n)] for i, a, n in zip(rng.integers(0, len(prot), 40_000), rng.integers(0, 20, 40_000), rng.integers(5, 40, 40_000))]
    pep = sorted(set(pep))
    vx = Index.build_var(pep, rng.choice(w.ids, len(pep)).astype(np.uint32))
    pick, miss = rng.integers(0, len(pep), 2_500_000).tolist(), (rng.random(2_500_000) >= 0.7).tolist()
    q = [pep[i][::-1] + b"W" if m else pep[i] for i, m in zip(pick, miss)]             # (reversed + one residue: mostly absent)
    data, qo = api._concat(q)
    d_b, d_qo = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).cuda(), torch.from_numpy(qo.view(np.int64).copy()).cuda()
    wv = torch.full((len(q),), -1, dtype=torch.int32, device="cuda")
    vx.lookup_var_dev(d_b, d_qo, wv, miss_value=9)
    torch.cuda.synchronize()
    got = vx.lookup_var(q, miss_value=9)
    assert np.array_equal(np.asarray(got).view(np.int32), wv.cpu().numpy())


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[2] at full size: 50 M pairs (5 x 10 M), 8 G-key table, size-independent properties
# ---------------------------------------------------------------------------------------------
def test_walk_index_beyond_32_bit_entry_numbers(api, torch_cuda):
    """The walk-order index at the size class of the real 9-mer index: 4.6 G windows of a synthetic proteome -- more entries than the
    table's 32-bit values can number, so the table holds entry number >> 1 and the key's low bits pick the entry of the pair
    (walk_ref.hpp).  Answers of 50 M windows of reads cut from the proteome (3 % of the residues changed), of packed 9-mers, and of
    the fused pipeline on 1 M pairs: bit for bit those of the plain table built from the same proteome."""
    import ctypes as C
    import gc
    torch = torch_cuda
    from umgap_amd import _lib
    from umgap_amd.api import Index, _dev_ptr, _stream_ptr
    from umgap_amd.synth import SyntheticWorkload
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 250e9:
        pytest.skip("needs ~230 GB of free HBM")
    L = _lib.load()
    n_frag, flen, per = 110_000_000, 50, 42
    w = SyntheticWorkload(n_tax=200_000, n_keys=1000, build_index=False)
    prot, poffs, ptaxa = _proteome(torch, w, n_frag, flen)
    hw = C.c_void_p()
    _lib.check(L.umgap_index_build_walk9_dev(_dev_ptr(prot), _dev_ptr(poffs), _dev_ptr(ptaxa), n_frag, flen, n_frag * per, 0.5, 0, C.byref(hw)))
    walk = Index(hw.value)
    shift, entries, exc = walk.walk_info()
    assert walk.kind == 10 and entries > (1 << 32) and shift == 1, (walk.num_keys, entries, shift, exc)
    hp = C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(poffs), _dev_ptr(ptaxa), n_frag, flen, n_frag * per, 0.5, 0, C.byref(hp)))
    plain = Index(hp.value)
    assert plain.num_keys == walk.num_keys
    n_reads, F = 600_000, 2
    reads = torch.empty(n_reads * F * flen, dtype=torch.uint8, device="cuda")
    _lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 44, 0, n_reads, F, 30, _dev_ptr(reads), 0, _stream_ptr(None)))
    roffs = torch.arange(0, (n_reads * F + 1) * flen, flen, dtype=torch.int64, device="cuda")
    oo = torch.arange(0, (n_reads * F + 1) * per, per, dtype=torch.int64, device="cuda")
    h0 = torch.full((n_reads * F * per,), -3, dtype=torch.int32, device="cuda")
    h1 = torch.full_like(h0, -5)
    plain.lookup_prot9_dev(reads, roffs, oo, h0, miss_value=0)
    walk.lookup_prot9_dev(reads, roffs, oo, h1, miss_value=0)
    torch.cuda.synchronize()
    assert torch.equal(h0, h1) and (h0 != 0).float().mean().item() > 0.5
    # packed 9-mers of the proteome's LAST fragments (entry numbers beyond 2^32) and random ones
    tail = prot[-flen * 2000:].cpu().numpy().reshape(2000, flen)
    pk, pok = _pack_windows(tail)
    q = np.concatenate([pk.ravel()[pok.ravel()], np.random.default_rng(3).integers(0, 1 << 45, 50_000, dtype=np.uint64)])
    d_q = torch.from_numpy(q.view(np.int64)).cuda()
    a, b = plain.lookup9_dev(d_q, miss_value=7), walk.lookup9_dev(d_q, miss_value=7)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and (a != 7).float().mean().item() > 0.5
    for method, strategy, lb, se in (("tree", "lca*", 2.0, True), ("rmq", "mrtl", 1.0, False)):
        outs = []
        for ix in (plain, walk):
            t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
            st = torch.full_like(t, -1)
            api.classify_prot9_dev(ix, w.taxonomy, reads, roffs, n_reads, F, t, st, method=method, strategy=strategy, factor=0.25, lower_bound=lb, seedextend=se,
                                   min_seed_size=3, max_gap_size=1)
            outs.append((t, st))
        torch.cuda.synchronize()
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (method, strategy)
    walk.close()
    plain.close()
    del prot, poffs, ptaxa, reads, h0, h1
    gc.collect()
    torch.cuda.empty_cache()


def test_walk_index_of_a_redundant_proteome(api, torch_cuda):
    """umgap_index_build_walk9_dev on a proteome whose fragments repeat (five copies of every fragment under different taxa, the
    copies far apart): expected_keys names the DISTINCT 9-mers, a fifth of the windows.  Every key keeps the smallest taxon of the
    fragments that hold it (the last pass of the values-free builder) and the kept repeats carry it too: answers as the plain table's."""
    import ctypes as C
    torch = torch_cuda
    from umgap_amd import _lib
    from umgap_amd.api import Index, _dev_ptr
    L = _lib.load()
    rng = np.random.default_rng(41)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    flen, n_base = 40, 6000
    base = rng.choice(letters, (n_base, flen))
    order = np.concatenate([rng.permutation(n_base) for _ in range(5)])
    frags = base[order]
    keys, ok = _pack_windows(base)
    distinct = int(len(np.unique(keys[ok])))
    n_frag = len(frags)
    ptaxa = rng.integers(2, 3000, n_frag).astype(np.uint32)
    d_p = torch.from_numpy(frags.reshape(-1).copy()).cuda()
    d_po = torch.arange(0, (n_frag + 1) * flen, flen, dtype=torch.int64, device="cuda")
    d_pt = torch.from_numpy(ptaxa.view(np.int32)).cuda()
    hp, hw = C.c_void_p(), C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(d_p), _dev_ptr(d_po), _dev_ptr(d_pt), n_frag, flen, distinct, 0.0, 0, C.byref(hp)))
    _lib.check(L.umgap_index_build_walk9_dev(_dev_ptr(d_p), _dev_ptr(d_po), _dev_ptr(d_pt), n_frag, flen, distinct, 0.0, 0, C.byref(hw)))
    plain, walk = Index(hp.value), Index(hw.value)
    q = np.concatenate([keys[ok].ravel(), rng.integers(0, 1 << 45, 20_000, dtype=np.uint64)])
    d_q = torch.from_numpy(q.view(np.int64)).cuda()
    a, b = plain.lookup9_dev(d_q, miss_value=0), walk.lookup9_dev(d_q, miss_value=0)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and (a != 0).float().mean().item() > 0.8
    # the smallest taxon over the copies, from first principles
    want = {}
    kk, oo = _pack_windows(frags)
    for f in range(n_frag):
        for k in kk[f][oo[f]]:
            want[int(k)] = min(want.get(int(k), 1 << 32), int(ptaxa[f]))
    got = a.cpu().numpy().view(np.uint32)[:int(ok.sum())]
    assert got.tolist() == [want[int(k)] for k in keys[ok].ravel()]
    # reads that follow the proteome (walks through kept repeats) answer the same through both
    n_reads = 3000
    ridx = rng.integers(0, n_frag, n_reads)
    d_r = torch.from_numpy(frags[ridx].reshape(-1).copy()).cuda()
    d_ro = torch.arange(0, (n_reads + 1) * flen, flen, dtype=torch.int64, device="cuda")
    d_oo = torch.arange(0, (n_reads + 1) * (flen - 8), flen - 8, dtype=torch.int64, device="cuda")
    h0 = torch.full((n_reads * (flen - 8),), -3, dtype=torch.int32, device="cuda")
    h1 = torch.full_like(h0, -5)
    plain.lookup_prot9_dev(d_r, d_ro, d_oo, h0, miss_value=0)
    walk.lookup_prot9_dev(d_r, d_ro, d_oo, h1, miss_value=0)
    torch.cuda.synchronize()
    assert torch.equal(h0, h1)
    walk.close()
    plain.close()


def test_walk_index_of_eight_billion_keys_builds_on_one_gpu(api, torch_cuda):
    """The walk-order index at the size of the real 9-mer index (BASELINE configs[1]: >= 8 G keys): 200 M fragments = 8.4 G windows
    through umgap_index_build_walk9_dev on the one GPU -- no values table beside the index under construction (round 4; that table
    was what did not fit), the values set by a last pass over the proteome.  The plain table of the same proteome answers a sample of
    reads first and is freed (both do not fit together); the walk index must give the same hits, window for window, and the same
    fused answers.  Entry numbers need 34 bits here: the table holds entry number >> 1."""
    import ctypes as C
    import gc
    torch = torch_cuda
    from umgap_amd import _lib
    from umgap_amd.api import Index, _dev_ptr, _stream_ptr
    from umgap_amd.synth import SyntheticWorkload
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 270e9:
        pytest.skip("needs ~250 GB of free HBM")
    L = _lib.load()
    n_frag, flen, per = 200_000_000, 50, 42
    w = SyntheticWorkload(n_tax=200_000, n_keys=1000, build_index=False)
    prot, poffs, ptaxa = _proteome(torch, w, n_frag, flen)
    n_reads, F = 400_000, 2
    reads = torch.empty(n_reads * F * flen, dtype=torch.uint8, device="cuda")
    _lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 45, 0, n_reads, F, 30, _dev_ptr(reads), 0, _stream_ptr(None)))
    roffs = torch.arange(0, (n_reads * F + 1) * flen, flen, dtype=torch.int64, device="cuda")
    oo = torch.arange(0, (n_reads * F + 1) * per, per, dtype=torch.int64, device="cuda")
    tail = prot[-flen * 2000:].cpu().numpy().reshape(2000, flen)        # the proteome's LAST fragments: the highest entry numbers
    pk, pok = _pack_windows(tail)
    d_q = torch.from_numpy(np.concatenate([pk.ravel()[pok.ravel()], np.random.default_rng(3).integers(0, 1 << 45, 50_000, dtype=np.uint64)]).view(np.int64)).cuda()

    def answers(ix):
        h = torch.full((n_reads * F * per,), -3, dtype=torch.int32, device="cuda")
        ix.lookup_prot9_dev(reads, roffs, oo, h, miss_value=0)
        a = ix.lookup9_dev(d_q, miss_value=7).clone()
        t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
        st = torch.full_like(t, -1)
        api.classify_prot9_dev(ix, w.taxonomy, reads, roffs, n_reads, F, t, st, method="tree", strategy="lca*", lower_bound=2.0, seedextend=True, min_seed_size=3, max_gap_size=1)
        torch.cuda.synchronize()
        return h, a, t, st

    hp = C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(poffs), _dev_ptr(ptaxa), n_frag, flen, n_frag * per, 0.5, 0, C.byref(hp)))
    plain = Index(hp.value)
    n_keys = plain.num_keys
    want = answers(plain)
    plain.close()
    gc.collect()
    torch.cuda.empty_cache()
    hw = C.c_void_p()
    _lib.check(L.umgap_index_build_walk9_dev(_dev_ptr(prot), _dev_ptr(poffs), _dev_ptr(ptaxa), n_frag, flen, n_frag * per, 0.5, 0, C.byref(hw)))
    walk = Index(hw.value)
    shift, entries, exc = walk.walk_info()
    assert walk.kind == 10 and walk.num_keys == n_keys and n_keys > 8_000_000_000 and entries > 8_000_000_000 and shift >= 1, (n_keys, walk.num_keys, entries, shift, exc)
    got = answers(walk)
    assert torch.equal(want[0], got[0]) and (got[0] != 0).float().mean().item() > 0.5
    assert torch.equal(want[1], got[1]) and (got[1] != 7).float().mean().item() > 0.5
    assert torch.equal(want[2], got[2]) and torch.equal(want[3], got[3])
    walk.close()
    del prot, poffs, ptaxa, reads
    gc.collect()
    torch.cuda.empty_cache()


def test_walk_index_of_eight_billion_keys_with_values_of_its_own(api, torch_cuda):
    """VERDICT r4 missing #3: the walk-order index at the size of the real 9-mer index WITH VALUES THAT DO NOT COME FROM THE PROTEOME
    (the real ninemer index holds the LCA values of joinkmers, /root/reference/README.md:54-57, not "the smallest taxon over the
    fragments").  umgap_walk_builder_new_deferred: the proteome first (200 M fragments = 8.4 G windows, sixteen chunks), then the
    (key, value) pairs streamed in 20 chunks -- value = a hash of the key, nothing a fragment could have told -- with no values table
    on the device at any time.  Checked on 400 k pairs of reads: the hits of every window (membership from the plain table of the same
    proteome, built first and freed; the value from the hash), get() of the proteome's last windows and of random keys, and the fused
    pipeline against seedextend + aggregate run on the expected hits."""
    import ctypes as C
    import gc
    torch = torch_cuda
    from umgap_amd import _lib
    from umgap_amd.api import Index, _dev_ptr, _stream_ptr
    from umgap_amd.synth import SyntheticWorkload
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 270e9:
        pytest.skip("needs ~260 GB of free HBM")
    L = _lib.load()
    n_frag, flen, per = 200_000_000, 50, 42
    w = SyntheticWorkload(n_tax=200_000, n_keys=1000, build_index=False)
    prot, poffs, ptaxa = _proteome(torch, w, n_frag, flen)
    n_reads, F = 400_000, 2
    reads = torch.empty(n_reads * F * flen, dtype=torch.uint8, device="cuda")
    _lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 45, 0, n_reads, F, 30, _dev_ptr(reads), 0, _stream_ptr(None)))
    roffs = torch.arange(0, (n_reads * F + 1) * flen, flen, dtype=torch.int64, device="cuda")
    oo = torch.arange(0, (n_reads * F + 1) * per, per, dtype=torch.int64, device="cuda")

    def window_keys(bytes2d):
        """uint8 [n, flen] on the device -> (packed 9-mers int64 [n, per], all nine letters in A..Z)"""
        code = bytes2d.to(torch.int64) - 65
        good = (code >= 0) & (code < 26)
        keys = torch.zeros((bytes2d.shape[0], per), dtype=torch.int64, device="cuda")
        ok = torch.ones((bytes2d.shape[0], per), dtype=torch.bool, device="cuda")
        for c in range(9):
            keys = (keys << 5) | (code[:, c:c + per] & 31)
            ok &= good[:, c:c + per]
        return keys, ok

    def value_of(keys):
        # one of four taxa, by a hash of the key (0x9E3779B97F4A7C15 as int64): nothing the fragment could have told, and few enough
        # for runs of equal hits (seeds) and repeated taxa (-l 2) to occur
        row = (((keys * -7046029254386353131) >> 20) & 3) * 1000 + 17
        return w.d_taxon_of_row[row]

    tail = prot[-flen * 2000:].view(2000, flen)
    tk, tok = window_keys(tail)
    d_q = torch.cat([tk[tok], torch.from_numpy(np.random.default_rng(3).integers(0, 1 << 45, 50_000, dtype=np.int64)).cuda()]).contiguous()
    # membership: the plain table of the proteome's windows (its own values are the fragments' taxa: only hit / miss is used)
    hp = C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(poffs), _dev_ptr(ptaxa), n_frag, flen, n_frag * per, 0.5, 0, C.byref(hp)))
    plain = Index(hp.value)
    n_keys = plain.num_keys
    member = torch.full((n_reads * F * per,), -3, dtype=torch.int32, device="cuda")
    plain.lookup_prot9_dev(reads, roffs, oo, member, miss_value=0)
    q_member = plain.lookup9_dev(d_q, miss_value=0).clone()
    torch.cuda.synchronize()
    plain.close()
    gc.collect()
    torch.cuda.empty_cache()
    rk, rok = window_keys(reads.view(n_reads * F, flen))
    want_hits = torch.where((member.view(-1, per) != 0) & rok, value_of(rk), torch.zeros_like(rk, dtype=torch.int32)).to(torch.int32).reshape(-1).contiguous()
    want_q = torch.where(q_member != 0, value_of(d_q), torch.full_like(q_member, 7)).to(torch.int32)
    assert (want_hits != 0).float().mean().item() > 0.5 and (want_q != 7).float().mean().item() > 0.5
    del member, rk, rok, q_member
    # the fused answers those hits must give: seedextend -g1 -s3 per read end, taxa2agg -l2 -a lca* (the unfused kernels, index-free)
    exp_hits = want_hits.clone()
    api.seedextend_dev(exp_hits, stride=per, n_records=n_reads * F, min_seed_size=3, max_gap_size=1)
    want_t = torch.empty(n_reads, dtype=torch.int32, device="cuda")
    want_s = torch.empty_like(want_t)
    w.taxonomy.aggregate_strided_dev(exp_hits, per * F, n_reads, want_t, want_s, method="tree", strategy="lca*", lower_bound=2.0)
    torch.cuda.synchronize()
    del exp_hits
    # the walk index: order first, values afterwards
    b = C.c_void_p()
    _lib.check(L.umgap_walk_builder_new_deferred(n_frag * per, 0.5, 0, C.byref(b)))
    try:
        n_chunks = 16
        for c in range(n_chunks):
            lo, hi = n_frag * c // n_chunks, n_frag * (c + 1) // n_chunks
            offs_c = (poffs[lo:hi + 1] - poffs[lo]).contiguous()
            _lib.check(L.umgap_walk_builder_add_dev(b, _dev_ptr(prot[lo * flen:]), _dev_ptr(offs_c), hi - lo, flen))
        n_vchunks = 20
        for c in range(n_vchunks):
            lo, hi = n_frag * c // n_vchunks, n_frag * (c + 1) // n_vchunks
            k, ok = window_keys(prot[lo * flen:hi * flen].view(hi - lo, flen))
            k = k[ok].contiguous()
            v = value_of(k).contiguous()
            _lib.check(L.umgap_walk_builder_set_values_dev(b, _dev_ptr(k), _dev_ptr(v), k.numel()))
            del k, v, ok
        hw = C.c_void_p()
        _lib.check(L.umgap_walk_builder_finish(b, C.byref(hw)))
        b = None
    finally:
        if b is not None:
            L.umgap_walk_builder_free(b)
    walk = Index(hw.value)
    shift, entries, exc = walk.walk_info()
    # (the plain builder reports the windows it was announced, this one the distinct keys it valued: 8.4 G random 9-mers collide 0.8 % of the time)
    assert walk.kind == 10 and 0.98 * n_keys < walk.num_keys <= n_keys and walk.num_keys > 8_000_000_000 and entries > 8_000_000_000 and shift >= 1, (n_keys, walk.num_keys, entries, shift, exc)
    got_hits = torch.full((n_reads * F * per,), -3, dtype=torch.int32, device="cuda")
    walk.lookup_prot9_dev(reads, roffs, oo, got_hits, miss_value=0)
    got_q = walk.lookup9_dev(d_q, miss_value=7)
    t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
    st = torch.full_like(t, -1)
    api.classify_prot9_dev(walk, w.taxonomy, reads, roffs, n_reads, F, t, st, method="tree", strategy="lca*", lower_bound=2.0, seedextend=True, min_seed_size=3, max_gap_size=1)
    torch.cuda.synchronize()
    assert torch.equal(got_hits, want_hits)
    assert torch.equal(got_q, want_q)
    assert torch.equal(st, want_s) and torch.equal(t, want_t) and (st == 0).float().mean().item() > 0.3
    walk.close()
    del prot, poffs, ptaxa, reads
    gc.collect()
    torch.cuda.empty_cache()


def test_config5_tryptic_index_one_billion_keys_sampled_vs_oracle(api, torch_cuda):
    """BASELINE.json configs[4] at size (SURVEY.md §8d Config 5; scripts/umgap-analyse.sh:289-298): a 1 G-key tryptic-peptide table
    (packed layout, 82 GB), 10 M pairs through umgap_classify_tryp_dev -- digest, length window 9..45, variable-length lookup,
    aggregate in one call -- in the script's presets: MRTL -l1 (tryptic-sensitivity), MRTL -l5 (tryptic-precision; on pairs of 2 x 4
    peptides, or nothing survives the filter) and tree hybrid f = 0.25 (the max-sensitivity aggregation).  Per mode 20 000 sampled pairs
    against the reference's digest restated (regex twice), the closed-form membership of the synthetic index and the oracle's
    literal aggregator with tie sets honoured (tests/tryptic_check.py); whole-batch properties beside it: the table bound to the
    taxonomy answers bit for bit like the unbound one, statuses are OK / EMPTY only, EMPTY prints the literal 1, answers are their
    own snap."""
    import gc
    import tryptic_check
    torch = torch_cuda
    from umgap_amd.synth import SyntheticWorkload, TrypticWorkload
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip("needs ~120 GB of free HBM")
    n_keys, n_pairs, F = 1_000_000_000, 10_000_000, 2
    w = SyntheticWorkload(n_tax=2_500_000, n_keys=1000, id_space=3_500_000, build_index=False)
    otx = tryptic_check.oracle_taxonomy(w)
    snap = w.taxonomy.snapping(False)
    tw = TrypticWorkload(n_keys, w, F, 2)
    ix, _ = tw.build_index()
    assert ix.var_layout == 1 and 60e9 < ix.table_bytes < 110e9
    rng = np.random.default_rng(55)
    d_t = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    d_s = torch.empty_like(d_t)
    d_t2, d_s2 = torch.empty_like(d_t), torch.empty_like(d_t)
    modes = (("rmq", "mrtl", 0.0, 1.0, 2, 0.9), ("rmq", "mrtl", 0.0, 5.0, 4, 0.2), ("tree", "hybrid", 0.25, 1.0, 2, 0.9))
    for k, (method, strategy, factor, lb, peps, min_ok) in enumerate(modes):
        twk = tw if peps == 2 else TrypticWorkload(n_keys, w, F, peps)
        d_bytes, d_foffs, _ = twk.batch(k, n_pairs)
        kw = dict(method=method, strategy=strategy, factor=factor, lower_bound=lb, min_len=9, max_len=45)
        n_pieces = api.classify_tryp_dev(ix, w.taxonomy, d_bytes, d_foffs, n_pairs, F, d_t, d_s, **kw)
        assert n_pieces == n_pairs * F * peps
        ix.bind(w.taxonomy)
        api.classify_tryp_dev(ix, w.taxonomy, d_bytes, d_foffs, n_pairs, F, d_t2, d_s2, **kw)
        torch.cuda.synchronize()
        ix.unbind(w.taxonomy)
        assert torch.equal(d_t, d_t2) and torch.equal(d_s, d_s2), (method, strategy, lb)
        t, s = d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32)
        assert ((s == 0) | (s == 1)).all() and (t[s == 1] == 1).all() and (snap[t] == t).all()
        assert (s == 0).mean() > min_ok, (method, strategy, lb, (s == 0).mean())
        sample = np.sort(rng.choice(n_pairs, 20_000, replace=False))
        rep = tryptic_check.check_sample(twk, k, n_pairs, sample, d_bytes.cpu().numpy(), d_foffs.cpu().numpy(), t, s, otx, method, strategy, factor, lb)
        assert rep["reads_checked"] == 20_000 and rep["pieces"] == 20_000 * F * peps and rep["member_pieces"] > 0.6 * rep["pieces"]
        assert rep["reads_with_an_answer"] > min_ok * 20_000
        del d_bytes, d_foffs
        torch.cuda.empty_cache()
    ix.close()
    w.taxonomy.close()
    torch.cuda.empty_cache()


FULL_N_TAX, FULL_N_KEYS = 2_500_000, 8_000_000_000


@pytest.fixture(scope="module")
def full_size(api, torch_cuda):
    """The benchmark's index and taxonomy as BASELINE.json names them: 8 G keys (128 GB of table), 2.5 M taxa.  Built once (1.1 s on the
    device) for the tests below, which are the last of the file; released when the module is through."""
    import gc
    torch = torch_cuda
    from umgap_amd.synth import SyntheticWorkload
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 200e9:
        pytest.skip("needs ~150 GB of free HBM")
    w = SyntheticWorkload(n_tax=FULL_N_TAX, n_keys=FULL_N_KEYS, id_space=int(FULL_N_TAX * 1.4), load_factor=0.5)
    assert 120e9 < w.index.table_bytes < 135e9          # 16 B/key at load 0.5: no grow-retries
    yield w
    w.index.close()
    w.taxonomy.close()
    torch.cuda.empty_cache()


def test_config2_one_million_pairs_full_index_surface_a(api, torch_cuda, full_size):
    """BASELINE.json configs[1] as written: 1 M synthetic pairs against the FULL 9-mer index (8 G keys), every read bit-exact vs the
    CPU oracle -- surface A (`pept2lca` form, packed 9-mers): all 84 M lookups against the closed-form index, every pair's LCA*
    (-l2) against the literal aggregator; then the fused kernel, unbound and bound to the taxonomy, bit for bit the same."""
    torch = torch_cuda
    from umgap_amd.synth import KMERS_PER_PAIR
    w, n_pairs = full_size, 1_000_000
    t, s = _pipeline_check(api, torch, FULL_N_TAX, FULL_N_KEYS, n_pairs, with_seedextend=False, w=w)
    assert (s == 0).mean() > 0.9
    d_keys = w.reads(0, n_pairs)
    d_t = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    d_s = torch.empty_like(d_t)
    for bound in (False, True):
        if bound:
            w.index.bind(w.taxonomy)
        d_t.fill_(-1)
        d_s.fill_(-1)
        try:
            api.classify9_dev(w.index, w.taxonomy, d_keys, n_pairs, KMERS_PER_PAIR, d_t, d_s, method="tree", lower_bound=2.0)
            torch.cuda.synchronize()
        finally:
            if bound:
                w.index.unbind(w.taxonomy)
        assert (d_t.cpu().numpy().view(np.uint32) == t).all() and (d_s.cpu().numpy().view(np.uint32) == s).all(), bound


def _fragments_from_kmers(torch, d_keys, n_pairs):
    """Surface B of the same reads (SURVEY.md 8d "two surfaces from the same data"): each end of a pair becomes ONE 50-residue
    fragment whose windows at offsets 0, 9, 18, 27 and 36 are the end's k-mers 0, 9, 18, 27, 36 spelled out (members of the index or
    not, as the generator chose them) and whose last 5 residues come from k-mer 41; the 37 windows in between straddle two k-mers
    and are whatever they are (1.6 % of random 9-mers are members of the 8 G-key index)."""
    k = d_keys.view(n_pairs * 2, 42)[:, [0, 9, 18, 27, 36, 41]]                          # [n_frag, 6]
    shifts = torch.arange(8, -1, -1, device=k.device, dtype=torch.int64) * 5
    codes = (k.unsqueeze(-1) >> shifts) & 31                                              # [n_frag, 6, 9]
    letters = (codes + 65).to(torch.uint8).reshape(n_pairs * 2, 54)
    return letters[:, :50].contiguous().reshape(-1)


def test_config2_one_million_pairs_full_index_surface_b(api, torch_cuda, full_size):
    """BASELINE.json configs[1], surface B (`prot2kmer2lca` form: one 50-residue line per end) against the same full index: the
    84 M windows of 1 M pairs -- umgap_index_lookup_prot9_dev -- against the closed-form oracle, and every pair's taxon (LCA*, -l2)
    from (i) the fused device kernel, unbound and bound, and (ii) umgap_classify_records fed from host arrays (what the command
    runs) against the literal aggregator."""
    torch = torch_cuda
    from umgap_amd.synth import ADD_B, MUL_A
    w, n_pairs, F, flen = full_size, 1_000_000, 2, 50
    per = flen - 8
    d_keys = w.reads(5_000_000, n_pairs)
    frags = _fragments_from_kmers(torch, d_keys, n_pairs)
    roffs = torch.arange(0, (n_pairs * F + 1) * flen, flen, dtype=torch.int64, device="cuda")
    # oracle: every window packed on the host, looked up in closed form
    host = frags.cpu().numpy()
    rk, rok = _pack_windows(host.reshape(n_pairs * F, flen))
    rk = rk.ravel()
    rk[~rok.ravel()] = np.uint64(1 << 45)                                                 # a byte outside 'A'..'Z': no key, a miss
    want_hits = O.synth_lookup9(O.SynthIndexParams(FULL_N_KEYS, FULL_N_TAX, MUL_A, ADD_B), w.taxon_of_row, rk, 0)
    designed = want_hits.reshape(n_pairs * F, per)[:, [0, 9, 18, 27, 36]]
    assert 0.55 < (designed != 0).mean() < 0.85 and 0.05 < (want_hits != 0).mean() < 0.2
    oo = torch.arange(0, (n_pairs * F + 1) * per, per, dtype=torch.int64, device="cuda")
    d_hits = torch.empty(n_pairs * F * per, dtype=torch.int32, device="cuda")
    w.index.lookup_prot9_dev(frags, roffs, oo, d_hits)
    torch.cuda.synchronize()
    assert (d_hits.cpu().numpy().view(np.uint32) == want_hits).all()
    del d_hits
    agg = O.Aggregator(O.Taxonomy.from_arrays(w.ids, w.parents, w.valid, w.rank), "tree", "lca*")
    offs = np.arange(n_pairs + 1, dtype=np.uint64) * (F * per)
    d_t = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    d_s = torch.empty_like(d_t)
    api.classify_prot9_dev(w.index, w.taxonomy, frags, roffs, n_pairs, F, d_t, d_s, method="tree", strategy="lca*", lower_bound=2.0)
    torch.cuda.synchronize()
    t, s = d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32)
    nbad, bad, wt, ws = agg.taxa2agg_check(want_hits, offs, t, s, lower_bound=2.0, nthreads=8)
    assert nbad == 0, (bad, t[bad.astype(int)], wt[bad.astype(int)])
    assert (s == 0).mean() > 0.3 and len(np.unique(t)) > 10_000
    w.index.bind(w.taxonomy)
    try:
        d_t2, d_s2 = torch.full_like(d_t, -1), torch.full_like(d_s, -1)
        api.classify_prot9_dev(w.index, w.taxonomy, frags, roffs, n_pairs, F, d_t2, d_s2, method="tree", strategy="lca*", lower_bound=2.0)
        torch.cuda.synchronize()
    finally:
        w.index.unbind(w.taxonomy)
    assert torch.equal(d_t, d_t2) and torch.equal(d_s, d_s2)
    # the host-pointer call of the command (umgap prot2kmer2agg): records in host memory in, one taxon per read in host memory out
    ht, hs = api.classify_records_arrays(w.index, w.taxonomy, host, roffs.cpu().numpy().view(np.uint64), np.arange(0, n_pairs * F + 1, F, dtype=np.uint64),
                                         method="tree", strategy="lca*", lower_bound=2.0, seedextend=False)
    assert (ht == t).all() and (hs == s).all()


def test_config2_one_million_pairs_proteome_index_plain_and_walk_order(api, torch_cuda):
    """Surface B over an index built from a proteome (umgap_index_build_windows9_dev: the smallest taxon over the fragments that hold
    a 9-mer), in the plain layout and in walk order: 1 M pairs of 2 x 50 residues cut from the proteome with 3 % of the residues
    changed, LCA* -l2, every pair against the oracle = a sorted array of the distinct 9-mers + the literal aggregator."""
    torch = torch_cuda
    import ctypes as C
    from umgap_amd import _lib
    from umgap_amd.api import Index, _dev_ptr, _stream_ptr
    from umgap_amd.synth import SyntheticWorkload
    L = _lib.load()
    n_tax, n_frag, flen, n_reads, F = 200_000, 500_000, 50, 1_000_000, 2
    per = flen - 8
    w = SyntheticWorkload(n_tax=n_tax, n_keys=1000, build_index=False)
    prot, poffs, ptaxa = _proteome(torch, w, n_frag, flen)
    reads = torch.empty(n_reads * F * flen, dtype=torch.uint8, device="cuda")
    _lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 44, 0, n_reads, F, 30, _dev_ptr(reads), 0, _stream_ptr(None)))
    roffs = torch.arange(0, (n_reads * F + 1) * flen, flen, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    pk, pok = _pack_windows(prot.cpu().numpy().reshape(n_frag, flen))
    assert pok.all()
    pt = np.repeat(ptaxa.cpu().numpy().view(np.uint32), per)
    order = np.lexsort((pt, pk.ravel()))
    ks, ts = pk.ravel()[order], pt[order]
    first = np.concatenate([[True], ks[1:] != ks[:-1]])
    sorted_index = O.SortedIndex(ks[first], ts[first])                                    # smallest taxon per distinct 9-mer
    del pk, pt, order, ks, ts, first
    rk, rok = _pack_windows(reads.cpu().numpy().reshape(n_reads * F, flen))
    rk = rk.ravel()
    rk[~rok.ravel()] = np.uint64(1 << 45)
    want_hits = sorted_index.lookup9(rk, 0, nthreads=8)
    assert 0.3 < (want_hits != 0).mean() < 0.95
    agg = O.Aggregator(O.Taxonomy.from_arrays(w.ids, w.parents, w.valid, w.rank), "tree", "lca*")
    offs = np.arange(n_reads + 1, dtype=np.uint64) * (F * per)
    got = {}
    for layout in ("plain", "walk order"):
        h = C.c_void_p()
        if layout == "plain":
            _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(poffs), _dev_ptr(ptaxa), n_frag, flen, n_frag * per, 0.0, 0, C.byref(h)))
        else:
            _lib.check(L.umgap_index_build_walk9_dev(_dev_ptr(prot), _dev_ptr(poffs), _dev_ptr(ptaxa), n_frag, flen, n_frag * per, 0.0, 0, C.byref(h)))
        ix = Index(h.value)
        assert ix.kind == (9 if layout == "plain" else 10)
        d_t = torch.empty(n_reads, dtype=torch.int32, device="cuda")
        d_s = torch.empty_like(d_t)
        api.classify_prot9_dev(ix, w.taxonomy, reads, roffs, n_reads, F, d_t, d_s, method="tree", strategy="lca*", lower_bound=2.0)
        torch.cuda.synchronize()
        t, s = d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32)
        nbad, bad, wt, ws = agg.taxa2agg_check(want_hits, offs, t, s, lower_bound=2.0, nthreads=8)
        assert nbad == 0, (layout, bad, t[bad.astype(int)], wt[bad.astype(int)])
        assert (s == 0).mean() > 0.5
        got[layout] = (t, s)
        ix.close()
    assert (got["plain"][0] == got["walk order"][0]).all() and (got["plain"][1] == got["walk order"][1]).all()


def test_full_size_50M_pairs_properties(api, torch_cuda, full_size):
    torch = torch_cuda
    from umgap_amd.synth import ADD_B, KMERS_PER_PAIR, MUL_A
    w = full_size
    n_tax, n_keys, B = FULL_N_TAX, FULL_N_KEYS, 10_000_000
    snap = w.taxonomy.snapping(False)
    ot = O.Taxonomy.from_arrays(w.ids, w.parents, w.valid, w.rank)
    agg = O.Aggregator(ot, "tree", "lca*")
    p = O.SynthIndexParams(n_keys, n_tax, MUL_A, ADD_B)
    rng = np.random.default_rng(50)
    d_t = torch.empty(B, dtype=torch.int32, device="cuda")
    d_s = torch.empty_like(d_t)
    d_t2, d_s2 = torch.empty_like(d_t), torch.empty_like(d_t)
    hist = np.zeros(4, np.int64)
    distinct = set()
    for b in range(5):
        keys = w.reads(b * B, B)
        api.classify9_dev(w.index, w.taxonomy, keys, B, KMERS_PER_PAIR, d_t, d_s, method="tree", lower_bound=2.0)
        # (1) fused == unfused, bit for bit, on the whole batch
        hits = w.index.lookup9_dev(keys)
        w.taxonomy.aggregate_strided_dev(hits, KMERS_PER_PAIR, B, d_t2, d_s2, method="tree", strategy="lca*", lower_bound=2.0)
        torch.cuda.synchronize()
        assert torch.equal(d_t, d_t2) and torch.equal(d_s, d_s2)
        t, s = d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32)
        # (2) output domain: status OK/EMPTY only; EMPTY prints the literal 1; every answer is its own snap (idempotent)
        hist += np.bincount(s, minlength=4)[:4]
        assert ((s == 0) | (s == 1)).all() and (t[s == 1] == 1).all()
        assert (snap[t] == t).all()
        distinct.update(np.unique(t[:200_000]).tolist())
        # (3) a random sample of reads against the oracle, end to end (closed-form index + literal LCA*)
        sample = np.sort(rng.choice(B, 20_000, replace=False))
        kh = keys.view(-1, KMERS_PER_PAIR)[torch.from_numpy(sample).cuda()].cpu().numpy().view(np.uint64).reshape(-1)
        want_hits = O.synth_lookup9(p, w.taxon_of_row, kh, 0)
        got_hits = hits.view(-1, KMERS_PER_PAIR)[torch.from_numpy(sample).cuda()].cpu().numpy().view(np.uint32).reshape(-1)
        assert (want_hits == got_hits).all()
        offs = np.arange(len(sample) + 1, dtype=np.uint64) * KMERS_PER_PAIR
        nbad, bad, _, _ = agg.taxa2agg_check(want_hits, offs, t[sample], s[sample], lower_bound=2.0, nthreads=8)
        assert nbad == 0, bad
        del keys, hits
    assert hist[0] > 0.9 * 5 * B and hist[2] == 0 and hist[3] == 0
    assert len(distinct) > 50_000      # the answers are spread over the taxonomy, not degenerate
    # (4) the table bound to the 2.5 M-taxon taxonomy (how bench.py runs the kernel): bit-identical on a whole batch
    keys = w.reads(4 * B, B)
    api.classify9_dev(w.index, w.taxonomy, keys, B, KMERS_PER_PAIR, d_t, d_s, method="tree", lower_bound=1.0, fragments_per_read=2, min_seed_size=3, max_gap_size=1)
    w.index.bind(w.taxonomy)
    api.classify9_dev(w.index, w.taxonomy, keys, B, KMERS_PER_PAIR, d_t2, d_s2, method="tree", lower_bound=1.0, fragments_per_read=2, min_seed_size=3, max_gap_size=1)
    torch.cuda.synchronize()
    w.index.unbind(w.taxonomy)
    assert torch.equal(d_t, d_t2) and torch.equal(d_s, d_s2)


def test_differential_fuzz_device_paths():
    """tools/fuzz_device.py, a short fixed-seed run: random taxonomies and reads (zeros, unknown ids, 0..200 hits) through the CSR,
    strided and fused device paths, every strategy, against the oracle's taxa2agg (a 1 500-iteration campaign over three seeds found
    nothing after the group aggregate's error-state guard went in)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([os.sys.executable, os.path.join(root, "tools", "fuzz_device.py"), "25", "7"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout[-3000:]

