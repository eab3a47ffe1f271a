"""Sampled end-to-end check of the fused tryptic pipeline (umgap_classify_tryp_dev) at full size -- test infrastructure, like
oracle_lib: used by tests/ and, outside its timed region, by bench.py's tryptic leg.

For a sample of read pairs of a TrypticWorkload batch: the read's bytes as they lie in device memory go through the REFERENCE's
digest restated literally (the regex `([KR])([^P])` -> "$1\\n$2" applied twice, `*` -> newline, empty lines dropped,
src/commands/prot2tryp2lca.rs:107-116) and its length window (:117-119); every piece is answered from the workload's closed form
(the host twin of the generator says which peptide number wrote those residues: a key iff number < n_keys, value = the taxon of row
number mod n_tax); the hits go through the oracle's literal aggregator (taxa2agg: count, -l filter, MRTL / hybrid / LCA*, snapping)
with tie sets honoured, and are compared with what the device call answered for that pair."""
import re

import numpy as np

import oracle_lib as O

_PAT = re.compile(rb"([KR])([^P])")


def tryptic_pieces(seq):
    once = _PAT.sub(rb"\1\n\2", seq)
    twice = _PAT.sub(rb"\1\n\2", once)
    return [p for p in twice.replace(b"*", b"\n").split(b"\n") if p]


def oracle_taxonomy(w):
    return O.Taxonomy.from_arrays(w.ids, w.parents, w.valid, w.rank)


def check_sample(tw, b, n_pairs, sample, host_bytes, host_foffs, got_taxon, got_status, otx, method, strategy, factor, lower_bound,
                 min_len=9, max_len=45, nthreads=8):
    """sample: sorted pair indices inside batch b.  host_bytes / host_foffs: the batch as the device holds it (numpy).  got_*: the
    device call's answers for the whole batch.  Returns a dict of what was checked; raises AssertionError on a difference."""
    F = tw.F
    ids = tw.piece_ids_np(b, n_pairs, sample)                       # (n, F * PEPS) peptide numbers by the generator's closed form
    flat = ids.reshape(-1)
    want_text = tw.peptides_np(flat)                                # ... and the residues the generator writes for them
    by_text = {}
    for t, i in zip(want_text, flat.tolist()):
        by_text.setdefault(t, i)
    values = dict(zip(by_text.values(), tw.values_np(np.array(list(by_text.values()), np.uint64)).tolist()))
    hits, roffs, n_pieces, n_members, n_window = [], [0], 0, 0, 0
    per = F * tw.PEPS
    for k, p in enumerate(sample.tolist()):
        got_pieces = []
        for f in range(F):
            lo, hi = int(host_foffs[p * F + f]), int(host_foffs[p * F + f + 1])
            got_pieces += tryptic_pieces(host_bytes[lo:hi].tobytes())
        # the digest of the bytes in device memory gives back exactly the peptides the closed form names (so the closed form speaks
        # about THESE bytes)
        assert got_pieces == want_text[k * per:(k + 1) * per], (p, got_pieces, want_text[k * per:(k + 1) * per])
        n_pieces += len(got_pieces)
        for piece in got_pieces:
            if min_len <= len(piece) <= max_len:
                n_window += 1
                v = values[by_text[piece]]
                n_members += v != 0
                hits.append(v)                                       # (without -o a miss prints nothing; 0 is what taxa2agg drops, :169)
        roffs.append(len(hits))
    agg = O.Aggregator(otx, method, strategy, factor)
    t = np.ascontiguousarray(got_taxon[sample]).view(np.uint32)
    s = np.ascontiguousarray(got_status[sample]).view(np.uint32)
    nbad, bad, _, _ = agg.taxa2agg_check(np.array(hits, np.uint32), np.array(roffs, np.uint64), t, s, lower_bound=lower_bound, nthreads=nthreads)
    assert nbad == 0, ("reads that differ from the oracle", [int(sample[i]) for i in bad[:10]], method, strategy, factor, lower_bound)
    return {"reads_checked": int(len(sample)), "pieces": n_pieces, "pieces_in_window": n_window, "member_pieces": int(n_members),
            "reads_with_an_answer": int((s == 0).sum()), "reads_empty": int((s == 1).sum()), "differences": 0}
