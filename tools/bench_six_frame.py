#!/usr/bin/env python3
"""Six-frame reads on the fused surface-B kernel: the two 9-mer SENSITIVITY pipelines of scripts/umgap-analyse.sh:277-288
feed `translate -a` output -- every pair as 12 protein fragments of ~50 residues, ~500 windows -- through
prot2kmer2lca | seedextend | uniq | taxa2agg.  Here: one launch of the LONG variant of k_classify_prot9 per batch
(umgap_classify_prot9_strategy_dev with 12 fragments per read).  Reads: frames 0 and 6 are proteome fragments with `mut` per mille
of their residues changed (the frames that code), the other ten are random residues.  Beside it the same index on the 2 x 50
gene-fragment form of the same pairs (84 windows per pair), for the cost per window.  Prints one JSON line.
    python tools/bench_six_frame.py [n_proteome_fragments] [n_pairs] [mutations_per_mille]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import _lib, synth
from umgap_amd.api import Index, _dev_ptr, _stream_ptr, classify_prot9_dev

L = _lib.load()
dev = torch.device("cuda", 0)
flen = 50
LEGS = (("max_sensitivity: seedextend -g1 -s2 | taxa2agg -l1 -m rmq -a mrtl (umgap-analyse.sh:277-282)", True, 2, "rmq", "mrtl", 1.0),
        ("high_sensitivity: seedextend -g1 -s3 | taxa2agg -l1 -a hybrid -f 0.25 (umgap-analyse.sh:283-288)", True, 3, "tree", "hybrid", 1.0),
        ("lca*: seedextend -g1 -s3 | taxa2agg -l2 -a lca*", True, 3, "tree", "lca*", 2.0))


def run(n_frag=50_000_000, n_pairs=2_000_000, mut=30, steps=3, workload=None):
    w = workload if workload is not None else synth.SyntheticWorkload(n_tax=2_500_000, n_keys=1000, build_index=False)
    offs = torch.arange(0, (n_frag + 1) * flen, flen, dtype=torch.int64, device=dev)
    prot = torch.empty(n_frag * flen, dtype=torch.uint8, device=dev)
    _lib.check(L.umgap_synth_peptides_dev(0, n_frag, 7, _dev_ptr(offs), _dev_ptr(prot), 0, _stream_ptr(None)))
    taxa = w.d_taxon_of_row[torch.arange(n_frag, device=dev) % w.n_tax].contiguous()
    h = C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * (flen - 8), 0.5, 0, C.byref(h)))
    ix = Index(h.value)
    del offs, taxa
    letters = torch.tensor(list(b"ACDEFGHIKLMNPQRSTVWY"), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev).manual_seed(6)
    two, twelve = [], []
    for b in range(steps):
        r = torch.empty(n_pairs * 2 * flen, dtype=torch.uint8, device=dev)
        _lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 44, b * n_pairs, n_pairs, 2, mut, _dev_ptr(r), 0, _stream_ptr(None)))
        frames = letters[torch.randint(0, 20, (n_pairs, 12, flen), device=dev, generator=g)]
        frames[:, 0, :] = r.view(n_pairs, 2, flen)[:, 0, :]
        frames[:, 6, :] = r.view(n_pairs, 2, flen)[:, 1, :]
        two.append(r)
        twelve.append(frames.reshape(-1).contiguous())
        del frames
    del prot
    ix.bind(w.taxonomy)
    d_t = torch.empty(n_pairs, dtype=torch.int32, device=dev)
    d_s = torch.empty_like(d_t)
    res = {"index_keys": n_frag * (flen - 8), "pairs_per_step": n_pairs, "mutations_per_mille": mut, "table_bound_to_taxonomy": True,
           "reads": "a pair = 12 fragments of 50 residues (504 windows): frames 0 and 6 proteome fragments with %d per mille of the residues changed, ten frames of random residues" % mut}
    for name, se, ms_, method, strategy, lb in LEGS:
        kw = dict(method=method, strategy=strategy, factor=0.25, lower_bound=lb, seedextend=se, min_seed_size=ms_, max_gap_size=1)
        out = {}
        for form, F, batches in (("six_frames", 12, twelve), ("gene_fragments", 2, two)):
            roffs = torch.arange(0, (n_pairs * F + 1) * flen, flen, dtype=torch.int64, device=dev)
            classify_prot9_dev(ix, w.taxonomy, batches[0], roffs, n_pairs, F, d_t, d_s, **kw)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(steps):
                classify_prot9_dev(ix, w.taxonomy, batches[i], roffs, n_pairs, F, d_t, d_s, **kw)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / steps
            st = d_s.cpu()
            out[form] = {"ms_per_step": round(ms, 3), "pairs_per_s": round(n_pairs / ms * 1e3, 1), "G_windows_per_s": round(n_pairs * F * 42 / ms / 1e6, 2),
                         "ns_per_window": round(ms * 1e6 / (n_pairs * F * 42), 4), "status_ok_frac": round(float((st == 0).float().mean()), 3),
                         "status_unsupported_frac": round(float((st == 4).float().mean()), 5), "taxon": d_t.clone(), "status": d_s.clone()}
            del roffs
        # the ten random frames add (almost) no hit: the six-frame answer is the gene-fragment answer wherever no random window hit
        same = float(((out["six_frames"]["taxon"] == out["gene_fragments"]["taxon"]) & (out["six_frames"]["status"] == out["gene_fragments"]["status"])).float().mean())
        for form in out:
            del out[form]["taxon"], out[form]["status"]
        out["same_answer_as_gene_fragments_frac"] = round(same, 4)
        out["cost_per_window_vs_gene_fragments"] = round(out["six_frames"]["ns_per_window"] / out["gene_fragments"]["ns_per_window"], 3)
        res[name] = out
    ix.unbind(w.taxonomy)
    ix.close()
    return res


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:4]]
    print(json.dumps(run(*a)))
