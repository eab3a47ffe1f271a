#!/usr/bin/env python3
"""Surface A (packed 9-mers in: what `pept2lca` gets from `prot2kmer`, /root/reference/src/commands/prot2kmer.rs: EVERY window of a
fragment, so a read's keys are shifts of each other) over the plain table and over the same proteome's index in WALK order through
umgap_classify9_dev (round 5: one probe per run of neighbouring 9-mers, the rest from the walk array; classify.hip
walk_group_to_strip).  Index: every 9-mer of a synthetic proteome; reads: proteome fragments with `mutations_per_mille` of their
residues replaced, packed into 84 keys per pair on the device.  The walk answers are checked bit for bit against the plain table's.
--pmc: one launch of each kernel after a warm-up (for `rocprofv3 --pmc TCC_EA0_RDREQ_sum`: requests per k-mer).  One JSON line.
    python tools/bench_surface_a_walk.py [n_proteome_fragments] [n_pairs] [mutations_per_mille]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from umgap_amd import _lib, synth
from umgap_amd.api import Index, _dev_ptr, _stream_ptr, classify9_dev

L = _lib.load()
dev = torch.device("cuda", 0)
flen, F, per = 50, 2, 42


def pack_windows(bytes2d):
    """uint8 [n, 50] on the device -> packed 9-mers int64 [n, 42]; a window with a byte outside A..Z gets a key that is no 9-mer"""
    code = bytes2d.to(torch.int64) - 65
    good = (code >= 0) & (code < 26)
    keys = torch.zeros((bytes2d.shape[0], per), dtype=torch.int64, device=dev)
    ok = torch.ones((bytes2d.shape[0], per), dtype=torch.bool, device=dev)
    for c in range(9):
        keys = (keys << 5) | (code[:, c:c + per] & 31)
        ok &= good[:, c:c + per]
    return torch.where(ok, keys, torch.full_like(keys, 1 << 50))


def run(n_frag=50_000_000, n_pairs=10_000_000, mut=30, steps=3, workload=None, pmc=False, phase_probe=False):
    w = workload if workload is not None else synth.SyntheticWorkload(n_tax=2_500_000, n_keys=1000, build_index=False)
    offs = torch.arange(0, (n_frag + 1) * flen, flen, dtype=torch.int64, device=dev)
    prot = torch.empty(n_frag * flen, dtype=torch.uint8, device=dev)
    _lib.check(L.umgap_synth_peptides_dev(0, n_frag, 7, _dev_ptr(offs), _dev_ptr(prot), 0, _stream_ptr(None)))
    taxa = w.d_taxon_of_row[torch.arange(n_frag, device=dev) % w.n_tax].contiguous()
    h, hw = C.c_void_p(), C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * per, 0.5, 0, C.byref(h)))
    _lib.check(L.umgap_index_build_walk9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * per, 0.5, 0, C.byref(hw)))
    plain, walk = Index(h.value), Index(hw.value)
    del offs, taxa
    batches = []
    for b in range(steps):
        r = torch.empty(n_pairs * F * flen, dtype=torch.uint8, device=dev)
        _lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 44, b * n_pairs, n_pairs, F, mut, _dev_ptr(r), 0, _stream_ptr(None)))
        keys = torch.empty(n_pairs * F * per, dtype=torch.int64, device=dev)
        step = 2_000_000                          # (fragments per packing pass: the intermediates are 8 bytes per residue)
        for lo in range(0, n_pairs * F, step):
            hi = min(n_pairs * F, lo + step)
            keys[lo * per:hi * per] = pack_windows(r[lo * flen:hi * flen].view(hi - lo, flen)).reshape(-1)
        batches.append(keys)
        del r
    del prot
    torch.cuda.empty_cache()
    d_t = torch.empty(n_pairs, dtype=torch.int32, device=dev)
    d_s = torch.empty_like(d_t)
    res = {"n_proteome_fragments": n_frag, "index_keys": n_frag * per, "pairs_per_step": n_pairs, "kmers_per_pair": F * per, "mutations_per_mille": mut,
           "plain_table_GB": round(plain.table_bytes / 1e9, 2), "walk_tables_GB": [round(walk.table_bytes / 1e9, 2), round(n_frag * per * 8 / 1e9, 2)],
           "reads": "pairs of 2 x 50 residues cut from the proteome, %.1f %% of the residues replaced, as 84 packed 9-mers (8 B each)" % (mut / 10.0)}
    legs = (("classify9", dict(method="tree", lower_bound=2.0)), ("classify9_seedextend", dict(method="tree", lower_bound=2.0, fragments_per_read=2, min_seed_size=3, max_gap_size=1)))
    want = {}
    for name, kw in legs:
        classify9_dev(plain, w.taxonomy, batches[0], n_pairs, F * per, d_t, d_s, **kw)
        torch.cuda.synchronize()
        want[name] = (d_t.clone(), d_s.clone())
    plain.bind(w.taxonomy)
    walk.bind(w.taxonomy)
    for layout, ix in (("plain_table", plain), ("walk_order_index", walk)):
        out = {}
        for name, kw in legs:
            classify9_dev(ix, w.taxonomy, batches[0], n_pairs, F * per, d_t, d_s, **kw)
            torch.cuda.synchronize()
            same = bool(torch.equal(d_t, want[name][0]) and torch.equal(d_s, want[name][1]))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            n_timed = 1 if pmc else 2 * steps
            for i in range(n_timed):
                classify9_dev(ix, w.taxonomy, batches[i % steps], n_pairs, F * per, d_t, d_s, **kw)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / n_timed
            out[name] = {"ms_per_step": round(ms, 3), "pairs_per_s": round(n_pairs / ms * 1e3, 1), "G_kmers_per_s": round(n_pairs * F * per / ms / 1e6, 2),
                         "bit_identical_to_unbound_plain_table": same, "status_ok_frac": round(float((d_s == 0).float().mean()), 3)}
        res[layout] = out
    if phase_probe:
        # where the walk kernel's time goes: the same launch with parts of walk_group_to_strip switched off (UMGAP_WALK_DEBUG bits: 1 no
        # phase B, 2 no entry loads, 4 no anchor probe, 8 no seedextend, 16 no aggregate; the answers are wrong then -- timing only)
        probe = {}
        for mode in (0, 1, 3, 7, 15, 23, 31, 16, 8):
            os.environ["UMGAP_WALK_DEBUG"] = str(mode)
            name, kw = legs[1]
            classify9_dev(walk, w.taxonomy, batches[0], n_pairs, F * per, d_t, d_s, **kw)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(2 * steps):
                classify9_dev(walk, w.taxonomy, batches[i % steps], n_pairs, F * per, d_t, d_s, **kw)
            b.record()
            torch.cuda.synchronize()
            probe[str(mode)] = round(a.elapsed_time(b) / (2 * steps), 3)
        del os.environ["UMGAP_WALK_DEBUG"]
        res["walk_phase_probe_ms"] = probe
    res["speedup_walk_over_plain"] = {name: round(res["plain_table"][name]["ms_per_step"] / res["walk_order_index"][name]["ms_per_step"], 2) for name, _ in legs}
    plain.unbind(w.taxonomy)
    walk.unbind(w.taxonomy)
    plain.close()
    walk.close()
    return res


if __name__ == "__main__":
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    print(json.dumps(run(int(argv[0]) if len(argv) > 0 else 50_000_000, int(argv[1]) if len(argv) > 1 else 10_000_000, int(argv[2]) if len(argv) > 2 else 30,
                         pmc="--pmc" in sys.argv, phase_probe="--phase-probe" in sys.argv)))
