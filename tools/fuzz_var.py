#!/usr/bin/env python3
"""Randomised stress of the byte-string (tryptic) index against a dict: both table layouts, key families that share all slots but
one, lengths around the slot boundaries, fills from 0.2 to 0.97 (displacement marks, stash), near-miss and malformed queries, save /
open / dump round trips.   python tools/fuzz_var.py [iterations] [seed]"""
import ctypes as C
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from umgap_amd import _lib, api

AZ = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ", np.uint8)
AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)


def rand_key(rng, alphabet, lo, hi):
    return rng.choice(alphabet, int(rng.integers(lo, hi + 1))).tobytes()


def one(rng, it):
    packed = bool(rng.integers(0, 4))              # 3 of 4 iterations: the packed layout
    os.environ["UMGAP_VAR_PACKED"] = "1" if packed else "0"
    alphabet = AZ if rng.integers(0, 2) else AA
    n = int(rng.choice([50, 3000, 60_000, 400_000]))
    lo, hi = [(1, 63), (5, 50), (14, 45), (15, 18), (38, 43), (60, 63), (1, 4)][int(rng.integers(0, 7))]
    keys = set()
    while len(keys) < n and not (hi <= 4 and len(keys) >= 20 ** hi // 2):
        k = rand_key(rng, alphabet, lo, hi)
        keys.add(k)
        if rng.random() < 0.3 and len(k) > 1:       # a family: same key, one residue / the length changed
            j = int(rng.integers(0, len(k)))
            keys.add(k[:j] + bytes([int(rng.choice(alphabet))]) + k[j + 1:])
            keys.add(k[:-1])
            if len(k) < 63:
                keys.add(k + b"A")
    raw = False
    if not packed and rng.random() < 0.5:          # byte strings outside 'A'..'Z' keep the legacy layout whatever the switch says
        for _ in range(max(1, n // 20)):
            keys.add(bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8)))
        raw = True
    keys = sorted(keys)
    vals = rng.integers(1, 1 << 32, len(keys), dtype=np.uint64).astype(np.uint32)
    d = dict(zip(keys, vals.tolist()))
    load = float(rng.choice([0.0, 0.2, 0.5, 0.8, 0.97]))
    ix = api.Index.build_var(keys, vals, load_factor=load)
    want_layout = 1 if (packed and not raw) else 0
    if packed and any(any(c < 65 or c > 90 for c in k) for k in keys):
        want_layout = 0
    assert ix.var_layout == want_layout, (ix.var_layout, want_layout)
    q = list(keys)
    q += [rand_key(rng, alphabet, 1, 63) for _ in range(min(20000, 2 * len(keys)))]
    q += [k[:-1] for k in keys[:5000] if len(k) > 1] + [k + b"A" for k in keys[:5000] if len(k) < 63] + [k[1:] for k in keys[:3000] if len(k) > 1]
    q += [b"", b"A" * 64, b"a", b"A*A", bytes([255, 0, 1]), keys[0].lower()]
    order = rng.permutation(len(q))
    q = [q[i] for i in order]
    miss = int(rng.choice([0, 7, 0xFFFFFFFF]))
    want = np.array([d.get(p, miss) for p in q], np.uint32)
    got = ix.lookup_var(q, miss_value=miss)
    bad = int((got != want).sum())
    if bad == 0 and rng.random() < 0.3:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "ix")
            ix.save(path)
            ix2 = api.Index.open(path)
            bad += int((ix2.lookup_var(q, miss_value=miss) != want).sum())
            n_out, hb = C.c_size_t(), C.c_uint64()
            cap = len(keys) + 8
            offs, lens, taxa = np.zeros(cap, np.uint64), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
            L = _lib.load()
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            _lib.check(L.umgap_index_dump_var(ix2._h, None, 0, p(offs), p(lens), p(taxa), cap, C.byref(n_out), C.byref(hb)))
            heap = np.zeros(hb.value + 1, np.uint8)
            _lib.check(L.umgap_index_dump_var(ix2._h, p(heap), hb.value, p(offs), p(lens), p(taxa), cap, C.byref(n_out), C.byref(hb)))
            dumped = {heap[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes(): int(taxa[i]) for i in range(n_out.value)}
            if dumped != d or n_out.value != len(keys):
                bad += 1
    if bad:
        print("MISMATCH iteration %d: %d wrong (packed=%s n=%d len %d..%d load %.2f)" % (it, bad, packed, len(keys), lo, hi, load), flush=True)
    return bad, len(q)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    bad = checks = 0
    for it in range(iters):
        b, c = one(rng, it)
        bad += b
        checks += c
    print("fuzz_var: %d iterations, %d lookups checked, %d mismatches (seed %d)" % (iters, checks, bad, seed))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
