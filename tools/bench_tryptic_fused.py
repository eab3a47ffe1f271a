#!/usr/bin/env python3
"""BASELINE.json configs[4] as ONE device call per batch: the tryptic presets of scripts/umgap-analyse.sh:289-298
(`prot2tryp2lca -l9 -L45 | uniq -d / | taxa2agg -l1|-l5 -m rmq -a mrtl`) through umgap_classify_tryp_dev: protein fragments in ->
digest on the device -> variable-length lookup (length window inside the probe) -> aggregate, one taxon per pair out.

Index: n_keys synthetic peptides that a tryptic digest gives back whole (umgap_synth_tryptic_dev: 14..45 residues, no K/R inside,
no leading P, K/R at the end).  A pair = 2 fragments (read ends) of 2 peptides each, written back to back; 70 % of the peptides are
members of the index, 30 % strangers; the members of a pair come from one source taxon's neighbourhood with probability
`related`.  Prints one JSON line.
    python tools/bench_tryptic_fused.py [n_keys] [n_pairs]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import _lib, synth
from umgap_amd.api import Index, _dev_ptr, _stream_ptr, classify_tryp_dev, digest_tryp_dev

F, PEPS, SEED = 2, 2, 43
L = _lib.load()
dev = torch.device("cuda", 0)


def splitmix(x):
    x = x + torch.tensor(-7046029254386353131, dtype=torch.int64, device=dev)  # 0x9E3779B97F4A7C15
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * torch.tensor(-4658895280553007687, dtype=torch.int64, device=dev)
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * torch.tensor(-7723592293110705685, dtype=torch.int64, device=dev)
    return x ^ ((x >> 31) & 0x1FFFFFFFF)


def lengths(ids):
    return 14 + (splitmix(ids ^ SEED) & 0x7FFFFFFF) % 32          # 14..45 residues


def write_peptides(ids, first, n):
    """bytes + offsets of peptides `ids` (or first .. first + n - 1), back to back"""
    idx = ids if ids is not None else torch.arange(first, first + n, dtype=torch.int64, device=dev)
    offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lengths(idx), 0, out=offs[1:])
    total = int(offs[-1].item())
    data = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    _lib.check(L.umgap_synth_tryptic_dev(_dev_ptr(ids) if ids is not None else None, first, n, SEED, _dev_ptr(offs), _dev_ptr(data), 0, _stream_ptr(None)))
    return data, offs, total


def run(n_keys=1_000_000_000, n_pairs=10_000_000, steps=3, workload=None):
    """Builds the index, the read batches, times the presets; returns the result dict.  `workload`: a SyntheticWorkload whose
    taxonomy to reuse (bench.py), else one is made."""
    w = workload if workload is not None else synth.SyntheticWorkload(n_tax=2_500_000, n_keys=1000, build_index=False)
    res = {"n_keys": n_keys, "pairs_per_step": n_pairs, "fragments_per_pair": F, "peptides_per_fragment": PEPS}
    t0 = time.perf_counter()
    data, offs, total = write_peptides(None, 0, n_keys)
    taxa = w.d_taxon_of_row[torch.arange(n_keys, device=dev) % w.n_tax].contiguous()
    h = C.c_void_p()
    _lib.check(L.umgap_index_build_var_dev(_dev_ptr(data), _dev_ptr(offs), _dev_ptr(taxa), n_keys, total, 0.0, 0, C.byref(h)))
    ix = Index(h.value)
    torch.cuda.synchronize()
    res["index_layout"] = "keys in the slots (no heap)" if ix.var_layout == 1 else "slots + heap"
    res["key_bytes_GB"] = round(total / 1e9, 2)
    res["table_GB"] = round(ix.table_bytes / 1e9, 2)
    res["setup_s"] = round(time.perf_counter() - t0, 1)
    del data, offs, taxa
    torch.cuda.empty_cache()

    # reads: pair p draws a source row s; each of its F * PEPS peptides is, with 70 %, a member whose taxon row is s (number = s + k *
    # n_tax: every n_tax-th peptide carries the same taxon) or, one time in four, any member; with 30 % a stranger (number >= 10^12)
    batches = []
    for b in range(steps):
        e = torch.arange(b * n_pairs * F * PEPS, (b + 1) * n_pairs * F * PEPS, dtype=torch.int64, device=dev)
        r = splitmix(e ^ 0x5DEECE66D)
        pair = e // (F * PEPS)
        src = (splitmix(pair ^ 0x2545F491) & 0x7FFFFFFFFFFF) % w.n_tax
        per_row = max(1, n_keys // w.n_tax)
        same = src + ((r >> 8) & 0x7FFFFFFF) % per_row * w.n_tax
        anym = ((r >> 8) & 0x7FFFFFFFFFFF) % n_keys
        cls = (r & 0xFF) % 100
        ids = torch.where(cls < 52, same, torch.where(cls < 70, anym, 10**12 + ((r >> 8) & 0x7FFFFFFFFFFF)))
        ids = torch.clamp(ids, min=0)
        d, o, tot = write_peptides(ids.contiguous(), 0, ids.numel())
        foffs = o[::PEPS].contiguous()                     # a fragment = PEPS peptides
        batches.append((d, foffs, tot))
        del e, r, pair, src, same, anym, cls, ids, o
    res["input_bytes_per_pair"] = round(batches[0][2] / n_pairs + 8 * F, 1)
    d_t = torch.empty(n_pairs, dtype=torch.int32, device=dev)
    d_s = torch.empty_like(d_t)
    # what the digest gives back: exactly F * PEPS pieces per pair
    po = digest_tryp_dev(batches[0][0], batches[0][1])
    assert po.numel() - 1 == n_pairs * F * PEPS, (po.numel() - 1, n_pairs * F * PEPS)
    del po
    # (tryptic-precision is the same call with -l5; a pair of this workload has four peptides, so it would only time empty reads)
    presets = (("tryptic_sensitivity_mrtl_l1", 1.0), ("tryptic_mrtl_l2", 2.0), ("max_sensitivity_tree_hybrid_f0.25_l1", 1.0))

    def time_presets(out, check=None):
        answers = {}
        for name, lb in presets:
            kw = dict(method="rmq", strategy="mrtl", lower_bound=lb, min_len=9, max_len=45)
            if "hybrid" in name:
                kw.update(method="tree", strategy="hybrid", factor=0.25)
            classify_tryp_dev(ix, w.taxonomy, batches[0][0], batches[0][1], n_pairs, F, d_t, d_s, **kw)
            torch.cuda.synchronize()
            answers[name] = (d_t.clone(), d_s.clone())
            per_step = []                         # every batch twice, each call timed on its own: the median is the figure (a call that has
            for i in range(2 * steps):            # to grow the handle's scratch -- a hipMalloc in a full GPU -- is an outlier, not the rate)
                t0 = time.perf_counter()
                n_pieces = classify_tryp_dev(ix, w.taxonomy, batches[i % steps][0], batches[i % steps][1], n_pairs, F, d_t, d_s, **kw)
                torch.cuda.synchronize()
                per_step.append((time.perf_counter() - t0) * 1e3)
            ms = sorted(per_step)[len(per_step) // 2]
            st = d_s.cpu()
            out[name] = {"pairs_per_s": round(n_pairs / ms * 1e3, 1), "ms_per_step": round(ms, 3), "ms_min_max": [round(min(per_step), 3), round(max(per_step), 3)], "pieces_per_step": n_pieces,
                         "G_pieces_per_s": round(n_pieces / ms / 1e6, 2), "status_ok_frac": round(float((st == 0).float().mean()), 3),
                         "status_empty_frac": round(float((st == 1).float().mean()), 3)}
            if check is not None:
                out[name]["bit_identical_to_unbound_table"] = bool(torch.equal(answers[name][0], check[name][0]) and torch.equal(answers[name][1], check[name][1]))
        return answers

    # the table as built (taxon ids in the slots: the aggregate gathers {pre, last} by id), then BOUND to the taxonomy
    # (umgap_index_bind_taxonomy: interval codes in the slots, no gather) -- the headline figures are the bound ones
    unbound = {}
    plain_answers = time_presets(unbound)
    t0 = time.perf_counter()
    ix.bind(w.taxonomy)
    torch.cuda.synchronize()
    res["bind_s"] = round(time.perf_counter() - t0, 2)
    res["table_bound_to_taxonomy"] = True
    time_presets(res, plain_answers)
    res["unbound_table"] = unbound
    ix.unbind(w.taxonomy)
    ix.close()
    return res


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000)))
