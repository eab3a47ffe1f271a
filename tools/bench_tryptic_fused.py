#!/usr/bin/env python3
"""BASELINE.json configs[4] as ONE device call per batch: the tryptic presets of scripts/umgap-analyse.sh:289-298
(`prot2tryp2lca -l9 -L45 | uniq -d / | taxa2agg -l1|-l5 -m rmq -a mrtl`) through umgap_classify_tryp_dev: protein fragments in ->
digest on the device -> variable-length lookup (length window inside the probe) -> aggregate, one taxon per pair out.

Index: n_keys synthetic peptides that a tryptic digest gives back whole (umgap_synth_tryptic_dev: 14..45 residues, no K/R inside,
no leading P, K/R at the end).  A pair = 2 fragments (read ends) of 2 peptides each, written back to back; 70 % of the peptides are
members of the index, 30 % strangers; the members of a pair come from one source taxon's neighbourhood with probability
`related`.  Prints one JSON line.
    python tools/bench_tryptic_fused.py [n_keys] [n_pairs]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from umgap_amd import synth
from umgap_amd.api import classify_tryp_dev, digest_tryp_dev

F, PEPS = 2, 2


def sampled_oracle_check(tw, b, n_pairs, batch, d_t, d_s, otx, kw, n_sample, seed):
    """`n_sample` pairs of batch b against the regex digest, the closed-form membership and the oracle's literal aggregator
    (tests/tryptic_check.py: the checker, outside every timed region)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tryptic_check
    sample = np.sort(np.random.default_rng(seed).choice(n_pairs, n_sample, replace=False))
    hb, hf = batch[0].cpu().numpy(), batch[1].cpu().numpy()
    return tryptic_check.check_sample(tw, b, n_pairs, sample, hb, hf, d_t.cpu().numpy(), d_s.cpu().numpy(), otx, kw["method"], kw["strategy"],
                                      kw.get("factor", 0.0), kw["lower_bound"], kw["min_len"], kw["max_len"])


def run(n_keys=1_000_000_000, n_pairs=10_000_000, steps=3, workload=None, oracle_sample=20_000):
    """Builds the index, the read batches, times the presets; returns the result dict.  `workload`: a SyntheticWorkload whose
    taxonomy to reuse (bench.py), else one is made.  oracle_sample: pairs per preset checked against the oracle (0: none)."""
    w = workload if workload is not None else synth.SyntheticWorkload(n_tax=2_500_000, n_keys=1000, build_index=False)
    tw = synth.TrypticWorkload(n_keys, w, F, PEPS)
    res = {"n_keys": n_keys, "pairs_per_step": n_pairs, "fragments_per_pair": F, "peptides_per_fragment": PEPS}
    t0 = time.perf_counter()
    ix, total = tw.build_index()
    res["index_layout"] = "keys in the slots (no heap)" if ix.var_layout == 1 else "slots + heap"
    res["key_bytes_GB"] = round(total / 1e9, 2)
    res["table_GB"] = round(ix.table_bytes / 1e9, 2)
    res["setup_s"] = round(time.perf_counter() - t0, 1)

    batches = [tw.batch(b, n_pairs) for b in range(steps)]
    res["input_bytes_per_pair"] = round(batches[0][2] / n_pairs + 8 * F, 1)
    d_t = torch.empty(n_pairs, dtype=torch.int32, device=w.device)
    d_s = torch.empty_like(d_t)
    # what the digest gives back: exactly F * PEPS pieces per pair
    po = digest_tryp_dev(batches[0][0], batches[0][1])
    assert po.numel() - 1 == n_pairs * F * PEPS, (po.numel() - 1, n_pairs * F * PEPS)
    del po
    presets = (("tryptic_sensitivity_mrtl_l1", 1.0), ("tryptic_mrtl_l2", 2.0), ("max_sensitivity_tree_hybrid_f0.25_l1", 1.0))
    otx = None
    if oracle_sample:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import tryptic_check
        otx = tryptic_check.oracle_taxonomy(w)

    def kw_of(name, lb):
        kw = dict(method="rmq", strategy="mrtl", lower_bound=lb, min_len=9, max_len=45)
        if "hybrid" in name:
            kw.update(method="tree", strategy="hybrid", factor=0.25)
        return kw

    def time_presets(out, check=None, oracle=False):
        answers = {}
        for name, lb in presets:
            kw = kw_of(name, lb)
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
            if oracle and otx is not None:
                # (the last timed call ran batch (2 * steps - 1) % steps: its answers are what d_t / d_s hold)
                b = (2 * steps - 1) % steps
                try:
                    out[name]["sampled_vs_oracle"] = sampled_oracle_check(tw, b, n_pairs, batches[b], d_t, d_s, otx, kw, oracle_sample, seed=len(name))
                except AssertionError as e:
                    out[name]["sampled_vs_oracle"] = {"differences": str(e)[:600]}
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
    time_presets(res, plain_answers, oracle=True)
    res["unbound_table"] = unbound
    if oracle_sample:
        # tryptic-precision (-l5, umgap-analyse.sh:289-292): a taxon must come up five times, so pairs of 2 x 4 peptides (one batch)
        tw8 = synth.TrypticWorkload(n_keys, w, F, 4)
        del batches
        torch.cuda.empty_cache()
        b8 = tw8.batch(7, n_pairs)
        kw = kw_of("tryptic_precision_mrtl_l5", 5.0)
        per_step = []
        for i in range(4):
            t0 = time.perf_counter()
            n_pieces = classify_tryp_dev(ix, w.taxonomy, b8[0], b8[1], n_pairs, F, d_t, d_s, **kw)
            torch.cuda.synchronize()
            per_step.append((time.perf_counter() - t0) * 1e3)
        ms = sorted(per_step[1:])[1]
        st = d_s.cpu()
        res["tryptic_precision_mrtl_l5_pairs_of_8_peptides"] = {"pairs_per_s": round(n_pairs / ms * 1e3, 1), "ms_per_step": round(ms, 3), "pieces_per_step": n_pieces,
                                                                "status_ok_frac": round(float((st == 0).float().mean()), 3)}
        try:
            res["tryptic_precision_mrtl_l5_pairs_of_8_peptides"]["sampled_vs_oracle"] = sampled_oracle_check(tw8, 7, n_pairs, b8, d_t, d_s, otx, kw, oracle_sample, seed=5)
        except AssertionError as e:
            res["tryptic_precision_mrtl_l5_pairs_of_8_peptides"]["sampled_vs_oracle"] = {"differences": str(e)[:600]}
    ix.unbind(w.taxonomy)
    ix.close()
    return res


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000)))
