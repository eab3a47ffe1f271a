#!/usr/bin/env python3
"""Randomised differential test of the DEVICE aggregation paths against the oracle (the CLI fuzzer, tools/fuzz_cli.py, feeds byte
soup through the commands; this one feeds taxon soup through the C ABI).  Every iteration draws a taxonomy (shape, invalid and
unranked nodes, ids missing from it), reads (lengths 0..200, zeros, unknown ids, few or many distinct taxa) and runs them
through
  * umgap_aggregate_dev            CSR: k_agg_lca / k_agg_csr_group (+ the large-scratch passes),
  * umgap_aggregate_strided_dev    fixed stride: k_agg_lca / k_agg_strided_group,
  * umgap_classify9[_strategy]_dev the fused kernels over an index whose values are those taxa (unknown ids included), with and
                                   without the table bound to the taxonomy where binding is possible,
  * umgap_digest_tryp_dev          byte soup against the reference's regex digest,
  * umgap_classify_prot9[_strategy]_dev, umgap_classify_tryp_dev   random proteins against a dict + the oracle,
  * the 9-mer index in walk order (umgap_index_build_walk9_dev)   against the plain table of the same random proteome, bit for bit,
  * umgap_sharded_{lookup9,classify9}_dev   N ranks as threads on this GPU against the whole-table kernels (every 4th iteration),
for every strategy, both methods, several -l / -f, ranked and unranked snapping, and checks each answer with the oracle's
taxa2agg (tie sets honoured).  Prints one line per mismatch and a summary; exit status 1 if anything differed.
    python tools/fuzz_device.py [iterations] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import oracle_lib as O
from umgap_amd import api
from umgap_amd.synth import synthetic_taxonomy

LETTERS = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
MODES = [("tree", "lca*"), ("rmq", "lca*"), ("tree", "hybrid"), ("rmq", "hybrid"), ("rmq", "mrtl"), ("tree", "mrtl")]


def draw_taxonomy(rng):
    n = int(rng.choice([30, 300, 5000]))
    ids, par, valid, rank = synthetic_taxonomy(n, int(n * rng.uniform(1.2, 3.0)), max_depth=int(rng.integers(3, min(30, max(4, n // 4)))),
                                               seed=int(rng.integers(1, 1 << 30)))
    return ids, par, valid, rank


def draw_reads(rng, ids, id_space, n_reads, max_len):
    unknown = np.setdiff1d(np.arange(1, id_space + 50, dtype=np.uint32), ids)[:200]
    reads = []
    for _ in range(n_reads):
        n = int(rng.integers(0, max_len + 1))
        style = rng.random()
        if style < 0.4:
            pool = rng.choice(ids, int(rng.integers(1, 6)))
        elif style < 0.8:
            pool = rng.choice(ids, int(rng.integers(1, 80)))
        else:
            pool = ids
        r = rng.choice(pool, n).astype(np.uint32)
        r[rng.random(n) < rng.choice([0.0, 0.3, 0.9])] = 0
        if len(unknown) and rng.random() < 0.15 and n:
            r[rng.integers(0, n, int(rng.integers(1, 4)))] = rng.choice(unknown)
        reads.append(r)
    return reads


def digest_leg(rng, fails):
    """umgap_digest_tryp_dev on byte soup (any byte but a newline, which a FASTA sequence cannot hold) against the reference's regex
    digest (src/commands/prot2tryp2lca.rs:107-116)."""
    import re
    pat = re.compile(rb"([KR])([^P])")
    alphabet = np.frombuffer(bytes(b for b in range(256) if b != 10) + b"KRP*" * 40 + b"ACDEFGHILMNQSTVWY" * 8, np.uint8)
    n_frag = int(rng.choice([1, 3, 50, 2000]))
    frags = [rng.choice(alphabet, int(rng.choice([0, 1, 2, 9, 60, 300, 5000], p=[.1, .1, .1, .2, .3, .15, .05]))).tobytes() for _ in range(n_frag)]
    pad = int(rng.integers(0, 20))
    data = np.frombuffer(b"K" * pad + b"".join(frags) + b"RR", np.uint8)
    offs = np.zeros(n_frag + 1, np.int64)
    offs[1:] = np.cumsum([len(f) for f in frags])
    got = api.digest_tryp_dev(torch.from_numpy(data.copy()).cuda(), torch.from_numpy(offs + pad).cuda()).cpu().numpy()
    host = data.tobytes()
    pieces = [host[int(a):int(b)] for a, b in zip(got[:-1], got[1:]) if host[int(a):int(b)] != b"*"]
    want = []
    for f in frags:
        twice = pat.sub(rb"\1\n\2", pat.sub(rb"\1\n\2", f))
        want += [p for p in twice.replace(b"*", b"\n").split(b"\n") if p]
    if pieces != want:
        fails.append("digest")
        print("MISMATCH digest n_frag", n_frag, "pieces", len(pieces), "want", len(want), flush=True)


def walk_leg(rng, fails):
    """The 9-mer index in walk order against the plain table of the same random proteome (ragged fragments, odd bytes, repeated
    fragments under other taxa): the fused surface-B pipeline (random strategy, fragments per read, seedextend) and the window get(),
    on reads that follow the proteome, deviate from it, or are random -- bit for bit."""
    import ctypes as C
    from umgap_amd import _lib
    from umgap_amd.api import Index, _dev_ptr
    L = _lib.load()
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    n_ids = int(rng.choice([15, 120, 900]))
    ids = np.arange(1, n_ids + 1, dtype=np.uint32)
    par = np.maximum(1, ids // int(rng.choice([2, 3, 5]))).astype(np.uint32)
    tx = api.Taxonomy(ids, par)
    frags = []
    for _ in range(int(rng.integers(5, 3000))):
        n = int(rng.choice([0, 4, 9, 10, 17, 50, 50, 64, 100, 130]))
        f = bytearray(rng.choice(letters[:int(rng.choice([3, 20]))], n).tobytes())     # a 3-letter alphabet makes duplicate 9-mers common
        if n > 9 and rng.random() < 0.1:
            f[int(rng.integers(0, n))] = int(rng.choice([42, 120, 64]))
        frags.append(bytes(f))
    frags += [frags[int(i)] for i in rng.integers(0, len(frags), len(frags) // 8)]
    n_frag = len(frags)
    n_win = int(sum(max(0, len(f) - 8) for f in frags))
    if n_win == 0:
        return
    data, offs = api._concat(frags)
    taxa = rng.choice(ids, n_frag).astype(np.uint32)
    d_p, d_po = torch.from_numpy(data.copy()).cuda(), torch.from_numpy(offs.view(np.int64)).cuda()
    d_pt = torch.from_numpy(taxa.view(np.int32)).cuda()
    max_len = max(len(f) for f in frags)
    hp, hw = C.c_void_p(), C.c_void_p()
    load = float(rng.choice([0.0, 0.3, 0.85]))
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(d_p), _dev_ptr(d_po), _dev_ptr(d_pt), n_frag, max_len, n_win, load, 0, C.byref(hp)))
    _lib.check(L.umgap_index_build_walk9_dev(_dev_ptr(d_p), _dev_ptr(d_po), _dev_ptr(d_pt), n_frag, max_len, n_win, load, 0, C.byref(hw)))
    plain, walk = Index(hp.value), Index(hw.value)
    F, n_reads = int(rng.choice([1, 2, 3, 7])), int(rng.integers(1, 600))
    mut = float(rng.choice([0.0, 0.01, 0.05, 0.3]))
    rfr = []
    for _ in range(n_reads * F):
        c = rng.random()
        if c < 0.75:
            f = bytearray(frags[int(rng.integers(0, n_frag))])
            for k in range(len(f)):
                if rng.random() < mut:
                    f[k] = int(rng.choice(letters))
            if len(f) > 12 and rng.random() < 0.3:                    # one residue lost or gained, or the read goes on in another fragment
                k, e = int(rng.integers(1, len(f) - 1)), rng.random()
                if e < 0.33:
                    del f[k]
                elif e < 0.66:
                    f.insert(k, int(rng.choice(letters)))
                else:
                    g = frags[int(rng.integers(0, n_frag))]
                    f = f[:k] + bytearray(g[min(k, len(g)):])
            a, b = (0, len(f)) if rng.random() < 0.5 else sorted(rng.integers(0, len(f) + 1, 2).tolist())
            rfr.append(bytes(f[a:b][:60]))
        else:
            rfr.append(rng.choice(letters, int(rng.integers(0, 60))).tobytes())
    rdata, roffs = api._concat(rfr)
    d_r = torch.from_numpy(rdata.copy()).cuda() if len(rdata) else torch.zeros(1, dtype=torch.uint8, device="cuda")
    d_ro = torch.from_numpy(roffs.view(np.int64)).cuda()
    oo = np.zeros(len(rfr) + 1, np.int64)
    oo[1:] = np.cumsum([max(0, len(f) - 8) for f in rfr])
    d_oo = torch.from_numpy(oo).cuda()
    if oo[-1]:
        h0 = torch.full((int(oo[-1]),), -3, dtype=torch.int32, device="cuda")
        h1 = torch.full_like(h0, -5)
        plain.lookup_prot9_dev(d_r, d_ro, d_oo, h0, miss_value=0)
        walk.lookup_prot9_dev(d_r, d_ro, d_oo, h1, miss_value=0)
        torch.cuda.synchronize()
        if not torch.equal(h0, h1):
            fails.append("walk get"); print("MISMATCH walk index: lookup_prot9 differs from the plain table", n_frag, n_reads, F, mut, flush=True)
    method, strategy = MODES[int(rng.integers(0, 5))]
    kw = dict(method=method, strategy=strategy, factor=float(rng.choice([0.1, 0.25, 0.6])), lower_bound=float(rng.choice([0.0, 1.0, 2.0])),
              seedextend=bool(rng.integers(0, 2)), min_seed_size=int(rng.choice([2, 3])), max_gap_size=int(rng.choice([0, 1])))
    outs = []
    for ix in (plain, walk):
        t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
        st = torch.full_like(t, -1)
        api.classify_prot9_dev(ix, tx, d_r, d_ro, n_reads, F, t, st, **kw)
        outs.append((t, st))
    torch.cuda.synchronize()
    outs_b = outs[0]
    if not (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])):
        fails.append("walk fused"); print("MISMATCH walk index: fused pipeline differs from the plain table", n_frag, n_reads, F, mut, kw, flush=True)
    # surface A (round 5): packed 9-mers of consecutive windows through umgap_classify9[_strategy]_dev on the walk index -- one probe
    # per run, the proteome's keys rebuilt along the entries -- against the plain table; reads of `wpr` windows cut from mutated
    # fragments, with and without the fragment structure, windows with a byte outside 'A'..'Z' packed as a non-key
    L = int(rng.choice([17, 29, 50]))
    long_frags = [f for f in frags if len(f) >= L]
    if long_frags:
        n2 = int(rng.integers(1, 400))
        F2 = int(rng.choice([1, 2, 3]))
        rows = np.empty((n2 * F2, L), np.uint8)
        for r in range(n2 * F2):
            if rng.random() < 0.85:
                g = long_frags[int(rng.integers(0, len(long_frags)))]
                a = int(rng.integers(0, len(g) - L + 1))
                f = bytearray(g[a:a + L])
                for k in range(L):
                    if rng.random() < mut:
                        f[k] = int(rng.choice(letters))
                if rng.random() < 0.2:       # chimera
                    g2 = long_frags[int(rng.integers(0, len(long_frags)))]
                    k = int(rng.integers(1, L - 1))
                    f = f[:k] + bytearray(g2[k:L])
                rows[r] = np.frombuffer(bytes(f), np.uint8)
            else:
                rows[r] = rng.choice(letters, L)
        code = rows.astype(np.int64) - 65
        good = (code >= 0) & (code < 26)
        W = L - 8
        keys = np.zeros((n2 * F2, W), np.uint64)
        ok = np.ones((n2 * F2, W), bool)
        for c in range(9):
            keys = (keys << np.uint64(5)) | (code[:, c:c + W] & 31).astype(np.uint64)
            ok &= good[:, c:c + W]
        keys = np.where(ok, keys, np.uint64(1 << 50)).reshape(-1)
        d_k = torch.from_numpy(keys.view(np.int64).copy()).cuda()
        stride = W * F2
        fr = int(rng.choice([0, F2]))
        se = fr > 0 and bool(rng.integers(0, 2))
        outs = []
        bind = bool(rng.integers(0, 2))
        for ix in (plain, walk):
            t = torch.full((n2,), -1, dtype=torch.int32, device="cuda")
            st = torch.full_like(t, -1)
            if bind:
                ix.bind(tx)
            api.classify9_dev(ix, tx, d_k, n2, stride, t, st, method=kw["method"], strategy=kw["strategy"], factor=kw["factor"], lower_bound=kw["lower_bound"],
                              fragments_per_read=fr, min_seed_size=2 if se else 1, max_gap_size=kw["max_gap_size"])
            torch.cuda.synchronize()
            if bind:
                ix.unbind(tx)
            outs.append((t, st))
        if not (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])):
            fails.append("walk surface A"); print("MISMATCH walk index: classify9 differs from the plain table", n_frag, n2, F2, L, mut, fr, se, bind, kw, flush=True)
        walk_leg.checks = getattr(walk_leg, "checks", 0) + n2
    # the deferred builder (round 5): the order from the proteome first, the plain table's (key, value) pairs streamed afterwards in
    # chunks of any order -> the same answers as the walk index built with its values
    pk, pv = plain.dump9()
    if len(pk):
        perm = rng.permutation(len(pk))
        pk, pv = pk[perm], pv[perm]
        # (the proteome in 1..3 chunks of whole fragments, the pairs in chunks of a random size)
        fcuts = sorted(set([0, n_frag] + rng.integers(0, n_frag + 1, int(rng.integers(0, 3))).tolist()))
        chunks = []
        for a, b in zip(fcuts[:-1], fcuts[1:]):
            cd, co = api._concat(frags[a:b])
            chunks.append((torch.from_numpy(cd.copy()).cuda() if len(cd) else torch.zeros(1, dtype=torch.uint8, device="cuda"), torch.from_numpy(co.view(np.int64)).cuda(),
                           max(1, max(len(f) for f in frags[a:b]))))
        step = int(rng.choice([7, 1000, 1 << 20]))
        vchunks = [(torch.from_numpy(pk[a:a + step].view(np.int64).copy()).cuda(), torch.from_numpy(pv[a:a + step].view(np.int32).copy()).cuda()) for a in range(0, len(pk), step)]
        wd = Index.build_walk_deferred(n_win, chunks, vchunks, load_factor=load)
        if oo[-1]:
            h2 = torch.full_like(h0, -7)
            wd.lookup_prot9_dev(d_r, d_ro, d_oo, h2, miss_value=0)
            torch.cuda.synchronize()
            if not torch.equal(h0, h2):
                fails.append("walk deferred"); print("MISMATCH walk index (deferred builder): lookup_prot9 differs from the plain table", n_frag, len(pk), flush=True)
        wd.close()
    # the record call with the residues in pieces (round 5), general and fused routes, plain and walk index
    if len(rdata) > 0 and n_reads:
        rf = np.arange(0, n_reads * F + 1, F, dtype=np.uint64)
        ncut = int(rng.integers(0, 6))
        edges = [0] + sorted(rng.integers(0, len(rdata) + 1, ncut).tolist()) + [len(rdata)]
        pieces = [rdata[a:b].copy() for a, b in zip(edges[:-1], edges[1:])]
        rkw = dict(method=kw["method"], strategy=kw["strategy"], factor=kw["factor"], lower_bound=kw["lower_bound"], seedextend=kw["seedextend"],
                   min_seed_size=kw["min_seed_size"], max_gap_size=kw["max_gap_size"])
        for ix in (plain, walk):
            ta, sa = api.classify_records_arrays(ix, tx, rdata, roffs, rf, **rkw)
            tb, sb = api.classify_records_pieces(ix, tx, pieces, roffs, rf, **rkw)
            if not ((ta == tb).all() and (sa == sb).all()):
                fails.append("records in pieces"); print("MISMATCH record call: pieces differ from one array", n_reads, F, edges, rkw, flush=True)
            # (a read the fused kernel does not take -- more windows than its strip holds -- is UMGAP_READ_UNSUPPORTED = 4 there; the record
            # call sends the whole call through its general kernels then)
            sd, td = outs_b[1].cpu().numpy().view(np.uint32), outs_b[0].cpu().numpy().view(np.uint32)
            took, ok_rows = sd != 4, sd == 0
            if not ((sa[took] == sd[took]).all() and (ta[ok_rows] == td[ok_rows]).all()):
                fails.append("records vs device-resident"); print("MISMATCH record call vs classify_prot9_dev", n_reads, F, rkw, flush=True)
    walk_leg.checks = getattr(walk_leg, "checks", 0) + n_reads + int(oo[-1])


def protein_legs(rng, fails):
    """The two fragment-fed fused pipelines on random proteins, own small taxonomy and indexes per call:
    umgap_classify_prot9[_strategy]_dev (every 9-residue window; ragged fragments, `*`, reads beyond the kernel's limits must be
    flagged, not guessed) and umgap_classify_tryp_dev (regex digest, length window), each against a dict + the oracle's taxa2agg."""
    import re
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    n_ids = int(rng.choice([15, 120]))
    ids = np.arange(1, n_ids + 1, dtype=np.uint32)
    par = np.maximum(1, ids // int(rng.choice([2, 3, 5]))).astype(np.uint32)
    tx = api.Taxonomy(ids, par)
    otx = O.Taxonomy.from_arrays(ids, par)
    method, strategy = MODES[int(rng.integers(0, 5))]
    factor, lb = float(rng.choice([0.1, 0.25, 0.6])), float(rng.choice([0.0, 1.0, 2.0]))
    agg = O.Aggregator(otx, method, strategy, factor)
    ctx = dict(method=method, strategy=strategy, lb=lb, f=factor, ranked=False)
    F, n_reads = int(rng.choice([1, 2, 3, 7])), int(rng.integers(20, 400))
    # ---- 9-mer windows ----
    kmers = sorted({rng.choice(letters, 9).tobytes() for _ in range(400)})
    kv = rng.choice(ids, len(kmers)).astype(np.uint32)
    kd = dict(zip(kmers, kv.tolist()))
    keys, _ = api.pack9_array(kmers)
    ix9 = api.Index.build(keys, kv)
    frags = []
    for _ in range(n_reads * F):
        n = int(rng.choice([0, 5, 9, 12, 30, 60, 140]))
        p = bytearray(rng.choice(letters, n).tobytes())
        for _k in range(n // 10):
            s0 = int(rng.integers(0, n - 8))
            p[s0:s0 + 9] = kmers[int(rng.integers(0, 30))]
        if n and rng.random() < 0.15:
            p[int(rng.integers(0, n))] = ord("*")
        frags.append(bytes(p))
    data, offs = api._concat(frags)
    d_data, d_offs = torch.from_numpy(data.copy()).cuda(), torch.from_numpy(offs.view(np.int64)).cuda()
    d_t = torch.full((n_reads,), -1, dtype=torch.int32, device="cuda")
    d_s = torch.full_like(d_t, -1)
    api.classify_prot9_dev(ix9, tx, d_data, d_offs, n_reads, F, d_t, d_s, method=method, strategy=strategy, factor=factor, lower_bound=lb)
    torch.cuda.synchronize()
    t, s = d_t.cpu().numpy().view(np.uint32).copy(), d_s.cpu().numpy().view(np.uint32).copy()
    hits, roffs, too_long = [], [0], np.zeros(n_reads, bool)
    for r in range(n_reads):
        fr = frags[r * F:(r + 1) * F]
        # reads of up to 4 fragments: 128 windows / 384 residues; of more (the LONG variant of the kernel): 512 / 1536, of which at most
        # 128 may carry a hit (prot_limits.hpp)
        lim_w, lim_b = (128, 384) if F <= 4 else (512, 1536)
        too_long[r] = sum(max(0, len(f) - 8) for f in fr) > lim_w or sum(len(f) for f in fr) > lim_b
        if not too_long[r]:
            mine = []
            for f in fr:
                mine += [kd.get(f[i:i + 9], 0) for i in range(max(0, len(f) - 8))]
            if F > 4 and sum(1 for h in mine if h) > 128:
                too_long[r] = True
            else:
                hits += mine
        roffs.append(len(hits))
    hits, roffs = np.array(hits, np.uint32), np.array(roffs, np.uint64)
    if not ((s[too_long] == 4).all() and (t[too_long] == 0).all()):
        fails.append("prot9 limits"); print("MISMATCH prot9: a read beyond the limits was not flagged", ctx, flush=True)
    unsupported = s == 4                                   # also: more than 64 distinct taxa after the filter (general strategies)
    _, _, wt, ws = agg.taxa2agg_check(hits, roffs, lower_bound=lb)
    if (unsupported & ~too_long).sum() > n_reads // 4:
        fails.append("prot9 unsupported"); print("MISMATCH prot9: too many reads refused", ctx, flush=True)
    t[unsupported], s[unsupported] = wt[unsupported], ws[unsupported]
    check("prot9 F=%d" % F, agg, hits, roffs, t, s, lb, fails, ctx)
    # ---- tryptic ----
    pat = re.compile(rb"([KR])([^P])")

    def peptide():
        k = int(rng.integers(1, 25))
        p = bytearray(rng.choice(np.frombuffer(b"ACDEFGHILMNPQSTVWY", np.uint8), k).tobytes())
        p[-1] = ord("K") if rng.random() < 0.5 else ord("R")
        if p[0] == ord("P"):
            p[0] = ord("A")
        return bytes(p)
    peps = sorted({peptide() for _ in range(800)})
    pv = rng.choice(ids, len(peps)).astype(np.uint32)
    ixv = api.Index.build_var(peps, pv)
    pd = dict(zip(peps, pv.tolist()))
    lo, hi = [(5, 50), (1, 63), (9, 45), (3, 7)][int(rng.integers(0, 4))]
    frags = []
    for _ in range(n_reads * F):
        parts = []
        for _k in range(int(rng.integers(0, 8))):
            u = rng.random()
            parts.append(peps[int(rng.integers(0, 40))] if u < 0.6 else (peptide() if u < 0.8 else (b"*" if u < 0.9 else rng.choice(letters, int(rng.integers(1, 80))).tobytes())))
        frags.append(b"".join(parts))
    data, offs = api._concat(frags)
    d_data, d_offs = torch.from_numpy(np.concatenate([data, np.zeros(1, np.uint8)])).cuda(), torch.from_numpy(offs.view(np.int64)).cuda()
    api.classify_tryp_dev(ixv, tx, d_data, d_offs, n_reads, F, d_t, d_s, method=method, strategy=strategy, factor=factor, lower_bound=lb, min_len=lo, max_len=hi)
    torch.cuda.synchronize()
    hits, roffs = [], [0]
    for r in range(n_reads):
        for f in frags[r * F:(r + 1) * F]:
            twice = pat.sub(rb"\1\n\2", pat.sub(rb"\1\n\2", f))
            hits += [pd.get(p, 0) for p in twice.replace(b"*", b"\n").split(b"\n") if p and lo <= len(p) <= hi]
        roffs.append(len(hits))
    check("tryp F=%d window %d..%d" % (F, lo, hi), agg, np.array(hits, np.uint32), np.array(roffs, np.uint64), d_t.cpu().numpy().view(np.uint32),
          d_s.cpu().numpy().view(np.uint32), lb, fails, ctx)


def sharded_leg(rng, fails):
    """The N-rank sharded path (pack -> exchange -> probe on the owner -> values back -> gather + aggregate) with all ranks as threads
    on this GPU (local transport): random world, pieces, stride, read counts and strategy; every rank's answers must be bit-identical
    to the whole-table kernels."""
    import threading

    from umgap_amd.dist import Comm, Sharded
    from umgap_amd.synth import ADD_B, MUL_A, SyntheticWorkload
    world, pieces = int(rng.choice([1, 2, 3, 5, 8])), int(rng.integers(1, 10))
    stride = int(rng.choice([6, 16, 42, 84, 128]))
    os.environ["UMGAP_SHARD_NO_LOCAL"] = str(int(rng.integers(0, 2)))
    n_keys = int(rng.choice([50_000, 600_000]))
    w = SyntheticWorkload(n_tax=int(rng.choice([500, 20_000])), n_keys=n_keys)
    n_reads = [int(rng.integers(1, 3000)) for _ in range(world)]
    max_keys = (max(n_reads) + pieces) // pieces * pieces * stride
    keys = [w.reads(int(rng.integers(0, 1 << 20)), n_reads[r], kmers_per_pair=stride) for r in range(world)]
    for k in keys:   # keys that are no packed 9-mers (a bit above 45 set on an otherwise present key): no owner, never sent, a miss
        bad = torch.rand(k.numel(), device="cuda") < 0.01
        k[bad] |= int(rng.integers(1, 1 << 18)) << 45
    method, strategy = MODES[int(rng.integers(0, 5))]
    lb, factor = float(rng.choice([0.0, 1.0, 2.0])), float(rng.choice([0.25, 0.6]))
    frags = int(rng.choice([f for f in (0, 1, 2, 3) if f == 0 or stride % f == 0]))
    kw = dict(method=method, strategy=strategy, factor=factor, lower_bound=lb, fragments_per_read=frags, min_seed_size=int(rng.integers(1, 4)), max_gap_size=int(rng.integers(0, 2)))
    want_hits, want = [], []
    for r in range(world):
        want_hits.append(w.index.lookup9_dev(keys[r], miss_value=0).clone())
        t = torch.empty(n_reads[r], dtype=torch.int32, device="cuda")
        s_ = torch.empty_like(t)
        api.classify9_dev(w.index, w.taxonomy, keys[r], n_reads[r], stride, t, s_, **kw)
        want.append((t, s_))
    torch.cuda.synchronize()
    shards = [api.Index.synthetic(n_keys, w.d_taxon_of_row, MUL_A, ADD_B, 0.0, r, world) for r in range(world)]
    comms = Comm.local(world)
    got_hits = [torch.full_like(want_hits[r], -7) for r in range(world)]
    got = [None] * world
    errs = [None] * world

    def body(r):
        try:
            torch.cuda.set_device(0)
            sh = Sharded(shards[r], w.taxonomy, comms[r], max_keys, pieces=pieces)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                sh.lookup9(keys[r], got_hits[r], miss_value=0, stream=st)
                t = torch.full((n_reads[r],), -1, dtype=torch.int32, device="cuda")
                s_ = torch.full_like(t, -1)
                sh.classify9(keys[r], n_reads[r], stride, t, s_, stream=st, **kw)
                got[r] = (t, s_)
            st.synchronize()
            sh.check()
            sh.close()
        except BaseException as e:  # noqa: BLE001
            errs[r] = e
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t_ in ts:
        t_.start()
    for t_ in ts:
        t_.join(120)
    ctx = dict(world=world, pieces=pieces, stride=stride, n_reads=n_reads, no_local=os.environ["UMGAP_SHARD_NO_LOCAL"], **kw)
    if any(t_.is_alive() for t_ in ts) or any(errs):
        fails.append("sharded")
        print("MISMATCH sharded: a rank failed or is stuck", ctx, [repr(e) for e in errs if e], flush=True)
        return
    torch.cuda.synchronize()
    for r in range(world):
        if not (torch.equal(got_hits[r], want_hits[r]) and torch.equal(got[r][0], want[r][0]) and torch.equal(got[r][1], want[r][1])):
            fails.append("sharded")
            print("MISMATCH sharded rank", r, ctx, flush=True)
            break
    for c in comms:
        c.close()


def seedextend_leg(rng, tx, agg, ix, d_keys, strided, lb, fails, ctx):
    """The fused probe -> seedextend -> aggregate kernels (every read end a fragment) against the oracle's literal seedextend
    state machine followed by its taxa2agg; a fragment on which the reference would panic (src/commands/seedextend.rs:169-173)
    must give READ_PANIC."""
    n, stride = strided.shape
    divisors = [f for f in (1, 2, 3, 4, 6, 8) if stride % f == 0 and stride // f >= 1]
    F = int(rng.choice(divisors))
    min_seed, max_gap = int(rng.choice([0, 1, 2, 3, 4])), int(rng.choice([0, 1, 2]))
    flen = stride // F
    hits, roffs, panicked = [], [0], np.zeros(n, bool)
    for r in range(n):
        for f in range(F):
            kept = O.seedextend_record(strided[r, f * flen:(f + 1) * flen], min_seed, max_gap)
            if kept is None:
                panicked[r] = True
            else:
                hits += kept.tolist()
        roffs.append(len(hits))
    d_t = torch.empty(n, dtype=torch.int32, device="cuda")
    d_s = torch.empty_like(d_t)
    api.classify9_dev(ix, tx, d_keys, n, stride, d_t, d_s, method=ctx["method"], strategy=ctx["strategy"], factor=ctx["f"], lower_bound=lb,
                      ranked_only=ctx["ranked"], fragments_per_read=F, min_seed_size=min_seed, max_gap_size=max_gap)
    torch.cuda.synchronize()
    t, s = d_t.cpu().numpy().view(np.uint32).copy(), d_s.cpu().numpy().view(np.uint32).copy()
    if not (s[panicked] == api.READ_PANIC).all():
        fails.append("seedextend panic")
        print("MISMATCH seedextend panic", ctx, dict(F=F, s=min_seed, g=max_gap), flush=True)
    hits, roffs = np.array(hits, np.uint32), np.array(roffs, np.uint64)
    _, _, wt, ws = agg.taxa2agg_check(hits, roffs, lower_bound=lb, ranked=ctx["ranked"])
    t[panicked], s[panicked] = wt[panicked], ws[panicked]          # those reads were checked above
    check("fused seedextend F=%d s=%d g=%d" % (F, min_seed, max_gap), agg, hits, roffs, t, s, lb, fails, ctx)


def check(tag, agg, taxa, offs, t, s, lb, fails, ctx):
    nbad, bad, wt, ws = agg.taxa2agg_check(taxa, offs, t, s, lower_bound=lb, ranked=ctx["ranked"])
    if nbad:
        b = int(bad[0])
        fails.append(tag)
        print("MISMATCH", tag, ctx, "read", b, "got", int(t[b]), int(s[b]), "want", int(wt[b]), int(ws[b]), "taxa", taxa[int(offs[b]):int(offs[b + 1])][:40].tolist(), flush=True)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    fails, n_checks = [], 0
    print("fuzz_device: seed", seed, flush=True)
    for it in range(iters):
        ids, par, valid, rank = draw_taxonomy(rng)
        tx = api.Taxonomy(ids, par, valid, rank)
        otx = O.Taxonomy.from_arrays(ids, par, valid, rank)
        max_len = int(rng.choice([6, 16, 84, 200]))
        reads = draw_reads(rng, ids, int(ids.max()), int(rng.integers(50, 1500)), max_len)
        taxa = np.concatenate(reads) if reads else np.zeros(0, np.uint32)
        offs = np.zeros(len(reads) + 1, np.uint64)
        offs[1:] = np.cumsum([len(r) for r in reads])
        d_taxa = torch.from_numpy(taxa.view(np.int32).copy()).cuda() if len(taxa) else torch.zeros(1, dtype=torch.int32, device="cuda")
        d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
        n = len(reads)
        d_t = torch.empty(n, dtype=torch.int32, device="cuda")
        d_s = torch.empty_like(d_t)
        # the fused kernels: an index whose keys are the (read, position) pairs and whose values are the taxa (zeros = absent keys)
        stride = max(1, max_len if max_len <= 128 else 128)
        fused_ok = max_len <= 128
        if fused_ok:
            strided = np.zeros((n, stride), np.uint32)
            for i, r in enumerate(reads):
                strided[i, :len(r)] = r
            d_strided = torch.from_numpy(strided.view(np.int32)).cuda()
            nz = np.flatnonzero(strided.ravel())
            keys_all = (np.arange(n * stride, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(1 << 44)
            keys_all = np.unique(keys_all, return_index=True)
            fused_ok = len(keys_all[0]) == n * stride
            if fused_ok:
                key_of = (np.arange(n * stride, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(1 << 44)
                ix = api.Index.build(key_of[nz], strided.ravel()[nz]) if len(nz) else None
                d_keys = torch.from_numpy(key_of.view(np.int64)).cuda()
        for method, strategy in MODES:
            for ranked in (False, True):
                lb = float(rng.choice([0.0, 1.0, 2.0, 3.0]))
                factor = float(rng.choice([0.1, 0.25, 0.5, 0.9]))
                if strategy == "mrtl" and method == "tree":
                    continue
                agg = O.Aggregator(otx, method, strategy, factor)
                ctx = dict(it=it, n_tax=len(ids), max_len=max_len, method=method, strategy=strategy, lb=lb, f=factor, ranked=ranked)
                kw = dict(method=method, strategy=strategy, factor=factor, lower_bound=lb, ranked_only=ranked)
                tx.aggregate_dev(d_taxa, d_offs, d_t, d_s, **kw)
                torch.cuda.synchronize()
                check("csr", agg, taxa, offs, d_t.cpu().numpy().view(np.uint32), d_s.cpu().numpy().view(np.uint32), lb, fails, ctx)
                n_checks += 1
                if max_len <= 128:
                    d_t2, d_s2 = torch.empty_like(d_t), torch.empty_like(d_s)
                    tx.aggregate_strided_dev(d_strided, stride, n, d_t2, d_s2, **kw)
                    torch.cuda.synchronize()
                    check("strided", agg, taxa, offs, d_t2.cpu().numpy().view(np.uint32), d_s2.cpu().numpy().view(np.uint32), lb, fails, ctx)
                    n_checks += 1
                    if fused_ok and ix is not None:
                        for bind in (False, True):
                            if bind:
                                try:
                                    ix.bind(tx)
                                except api.UmgapError:
                                    break          # a value outside the taxonomy: binding is refused, as documented
                            api.classify9_dev(ix, tx, d_keys, n, stride, d_t2, d_s2, method=method, strategy=strategy, factor=factor, lower_bound=lb,
                                              ranked_only=ranked)
                            torch.cuda.synchronize()
                            check("fused bound=%d" % bind, agg, taxa, offs, d_t2.cpu().numpy().view(np.uint32), d_s2.cpu().numpy().view(np.uint32), lb, fails, ctx)
                            n_checks += 1
                            if bind:
                                ix.unbind(tx)
                        seedextend_leg(rng, tx, agg, ix, d_keys, strided, lb, fails, ctx)
                        n_checks += 1
        digest_leg(rng, fails)
        protein_legs(rng, fails)
        walk_leg(rng, fails)
        n_checks += 2
        n_checks += 3
        if it % 4 == 0:
            sharded_leg(rng, fails)
            n_checks += 1
        if (it + 1) % 5 == 0:
            print("iteration %d: %d checks, %d mismatches" % (it + 1, n_checks, len(fails)), flush=True)
    print("fuzz_device: %d iterations, %d checks, %d mismatches (seed %d)" % (iters, n_checks, len(fails), seed))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
