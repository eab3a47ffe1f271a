"""Synthetic workload of the benchmark and the large parity tests (DESIGN.md "Synthetic workload";
plan: SURVEY.md §8d).  Everything large is generated on the GPU from seeds; only the taxonomy
(a few MB of numpy arrays) is made on the host.

  taxonomy (seed 42): n_tax ids drawn without replacement from [1, id_space], id 1 = root; nodes are
      placed on levels 1..max_depth with a bell-shaped level-size profile (mean depth ~ 0.6*max_depth)
      and take a uniformly random parent on the level above; ranks cycle through src/rank.rs:10-41;
      2 % of the taxa are invalid; rows are shuffled (children order = row order).
  index (seed 43): key_i = pi(i) = (a*i + b) mod 20^9 written base 20 over ACDEFGHIKLMNPQRSTVWY,
      i < n_keys; value = taxon of row (i mod n_tax).  Membership is closed-form for the oracle.
  reads (seed 44): per pair 84 packed 9-mers: 60 % members whose taxon lies on the root path of the
      pair's source taxon, 10 % unrelated members (pairs of slots share one, so some occur twice),
      30 % non-members (1 % of them with a letter outside the 20-letter alphabet).
"""
import numpy as np

SEED_TAXONOMY, SEED_INDEX, SEED_READS = 42, 43, 44
MUL_A = 22695477          # odd, not a multiple of 5, < 2^25  => pi is a bijection on [0, 20^9)
ADD_B = 314159265358      # < 20^9
M20_9 = 20 ** 9
KMERS_PER_PAIR = 84       # 2 fragments x (50 aa - 9 + 1) 9-mers


def synthetic_taxonomy(n_tax=2_500_000, id_space=3_500_000, max_depth=45, seed=SEED_TAXONOMY, invalid_frac=0.02):
    """Returns (ids, parents, valid, rank) uint arrays in row order; row 0 need not be the root."""
    assert 2 <= n_tax <= id_space
    rng = np.random.default_rng(seed)
    # level sizes: bell shape over 1..max_depth, at least 1 node per level, total n_tax-1
    levels = np.arange(1, max_depth + 1)
    w = np.exp(-0.5 * ((levels - 0.6 * max_depth) / (0.17 * max_depth)) ** 2) + 1e-4
    sizes = np.maximum(1, np.floor(w / w.sum() * (n_tax - 1)).astype(np.int64))
    if n_tax - 1 < max_depth:
        sizes = np.zeros(max_depth, np.int64)
        sizes[: n_tax - 1] = 1
    diff = (n_tax - 1) - sizes.sum()
    sizes[int(np.argmax(sizes))] += diff
    assert sizes.sum() == n_tax - 1 and sizes.min() >= 0
    ids = np.empty(n_tax, np.uint32)
    ids[0] = 1
    ids[1:] = rng.choice(np.arange(2, id_space + 1, dtype=np.uint32), n_tax - 1, replace=False)
    parents = np.empty(n_tax, np.uint32)
    parents[0] = 1
    lo_prev, hi_prev = 0, 1  # node index range of the previous level
    pos = 1
    for s in sizes:
        if s == 0:
            continue
        pick = rng.integers(lo_prev, hi_prev, size=s)
        parents[pos:pos + s] = ids[pick]
        lo_prev, hi_prev = pos, pos + s
        pos += s
    rank = (np.arange(n_tax) % 32).astype(np.uint8)
    rank[0] = 0
    valid = (rng.random(n_tax) >= invalid_frac).astype(np.uint8)
    valid[0] = 1
    perm = rng.permutation(n_tax)
    return ids[perm], parents[perm], valid[perm], rank[perm]


def write_taxonomy_tsv(path, ids, parents, valid, rank):
    """taxons.tsv in the exact row format of src/taxon.rs:65-114."""
    from .api import RANKS
    with open(path, "wb") as f:
        for i in range(len(ids)):
            f.write(b"%d\ttaxon%d\t%s\t%d\t%s\n" % (int(ids[i]), int(ids[i]), RANKS[int(rank[i])].encode(), int(parents[i]),
                                                    b"\x01" if valid[i] else b"\x00"))


def key_of_member(i, mul_a=MUL_A, add_b=ADD_B):
    """Host restatement of the device generator (for small tests)."""
    x = (mul_a * int(i) + add_b) % M20_9
    key = 0
    for j in range(9):
        d = x % 20
        x //= 20
        key |= (ord("ACDEFGHIKLMNPQRSTVWY"[d]) - 65) << (5 * j)
    return key


class SyntheticWorkload:
    """Taxonomy + index + read generator bound to one GPU."""

    def __init__(self, n_tax, n_keys, id_space=None, max_depth=45, load_factor=0.0, device=0, build_index=True, shard=0, n_shards=1):
        import torch
        from .api import Index, Taxonomy
        self.device = torch.device("cuda", device)
        self.n_tax, self.n_keys = int(n_tax), int(n_keys)
        id_space = id_space or max(int(n_tax * 1.4), n_tax + 1)
        self.ids, self.parents, self.valid, self.rank = synthetic_taxonomy(n_tax, id_space, max_depth)
        self.taxonomy = Taxonomy(self.ids, self.parents, self.valid, self.rank, device=device)
        self.taxon_of_row = self.ids.astype(np.uint32)
        row_of_id = np.full(len(self.taxonomy), 0xFFFFFFFF, np.uint32)
        row_of_id[self.ids] = np.arange(n_tax, dtype=np.uint32)
        self.d_taxon_of_row = torch.from_numpy(self.taxon_of_row.view(np.int32)).to(self.device)
        self.d_row_of_id = torch.from_numpy(row_of_id.view(np.int32)).to(self.device)
        self.index = Index.synthetic(self.n_keys, self.d_taxon_of_row, MUL_A, ADD_B, load_factor, shard, n_shards) if build_index else None

    def reads(self, first_pair, n_pairs, d_keys=None, kmers_per_pair=KMERS_PER_PAIR, stream=None, path_percent=60, unrelated_percent=10):
        """Packed 9-mers of pairs [first_pair, first_pair+n_pairs) as an int64 CUDA tensor (the 60 / 10 / 30 mix of the
        module docstring unless told otherwise)."""
        import ctypes as C
        import torch
        from . import _lib
        from .api import _dev_ptr, _stream_ptr
        if d_keys is None:
            d_keys = torch.empty(n_pairs * kmers_per_pair, dtype=torch.int64, device=self.device)
        _lib.check(_lib.load().umgap_synth_reads9_mix_dev(self.taxonomy._h, _dev_ptr(self.d_taxon_of_row), _dev_ptr(self.d_row_of_id),
                                                          self.n_tax, self.n_keys, MUL_A, ADD_B, SEED_READS, first_pair, n_pairs,
                                                          kmers_per_pair, path_percent, unrelated_percent, _dev_ptr(d_keys), _stream_ptr(stream)))
        return d_keys


# ---- tryptic workload (BASELINE.json configs[4]; SURVEY.md §8d Config 5) ---------------------------------------------------------
SEED_TRYPTIC = 43
_M64 = (1 << 64) - 1


def _splitmix_np(x):
    """splitmix64 on uint64 arrays (the device generator's mixer, common.hpp)."""
    with np.errstate(over="ignore"):
        x = x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


class TrypticWorkload:
    """Synthetic tryptic-peptide index + reads, generated on the device from closed forms that have host twins (numpy) so a test can
    say what any read holds without a table on the CPU.

    index: peptide number i < n_keys; 14..45 residues over ACDEFGHILMNPQSTVWY, no leading P, K or R at the end and nowhere else -- a
        tryptic digest gives it back whole (umgap_synth_tryptic_dev).  value = taxon of row (i mod n_tax).
    reads: pair p = `fragments` fragments (read ends) of `peptides` peptides written back to back.  A pair draws a source row s; each of
        its peptides is, 52 %: a member that carries s's taxon (number s + k * n_tax), 18 %: any member, 30 %: a stranger (number >=
        10^12: the same generator, so still a clean tryptic peptide, but no key)."""

    def __init__(self, n_keys, workload, fragments=2, peptides=2, seed=SEED_TRYPTIC):
        self.n_keys, self.w, self.F, self.PEPS, self.seed = int(n_keys), workload, int(fragments), int(peptides), int(seed)
        self.n_tax = workload.n_tax

    # -- device side ---------------------------------------------------------------------------------------------------------------
    def _splitmix_t(self, x):
        import torch
        dev = self.w.device
        x = x + torch.tensor(-7046029254386353131, dtype=torch.int64, device=dev)  # 0x9E3779B97F4A7C15
        x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * torch.tensor(-4658895280553007687, dtype=torch.int64, device=dev)
        x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * torch.tensor(-7723592293110705685, dtype=torch.int64, device=dev)
        return x ^ ((x >> 31) & 0x1FFFFFFFF)

    def _lengths_t(self, ids):
        return 14 + (self._splitmix_t(ids ^ self.seed) & 0x7FFFFFFF) % 32          # 14..45 residues

    def write_peptides(self, ids, first, n):
        """bytes + offsets of peptides `ids` (or first .. first + n - 1), back to back, on the device"""
        import torch
        from . import _lib
        from .api import _dev_ptr, _stream_ptr
        dev = self.w.device
        idx = ids if ids is not None else torch.arange(first, first + n, dtype=torch.int64, device=dev)
        offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        torch.cumsum(self._lengths_t(idx), 0, out=offs[1:])
        total = int(offs[-1].item())
        data = torch.empty(total + 64, dtype=torch.uint8, device=dev)
        _lib.check(_lib.load().umgap_synth_tryptic_dev(_dev_ptr(ids) if ids is not None else None, first, n, self.seed, _dev_ptr(offs), _dev_ptr(data),
                                                       dev.index or 0, _stream_ptr(None)))
        return data, offs, total

    def build_index(self):
        """The packed peptide table of all n_keys peptides; returns (Index, bytes of key text)."""
        import ctypes as C
        import torch
        from . import _lib
        from .api import Index, _dev_ptr
        data, offs, total = self.write_peptides(None, 0, self.n_keys)
        taxa = self.w.d_taxon_of_row[torch.arange(self.n_keys, device=self.w.device) % self.n_tax].contiguous()
        h = C.c_void_p()
        _lib.check(_lib.load().umgap_index_build_var_dev(_dev_ptr(data), _dev_ptr(offs), _dev_ptr(taxa), self.n_keys, total, 0.0, self.w.device.index or 0, C.byref(h)))
        ix = Index(h.value)
        torch.cuda.synchronize()
        del data, offs, taxa
        torch.cuda.empty_cache()
        return ix, total

    def batch(self, b, n_pairs):
        """Reads of pairs [b * n_pairs, (b + 1) * n_pairs): (bytes, fragment offsets (n_pairs * F + 1), total bytes), on the device."""
        import torch
        dev = self.w.device
        per = self.F * self.PEPS
        e = torch.arange(b * n_pairs * per, (b + 1) * n_pairs * per, dtype=torch.int64, device=dev)
        r = self._splitmix_t(e ^ 0x5DEECE66D)
        pair = e // per
        src = (self._splitmix_t(pair ^ 0x2545F491) & 0x7FFFFFFFFFFF) % self.n_tax
        per_row = max(1, self.n_keys // self.n_tax)
        same = src + ((r >> 8) & 0x7FFFFFFF) % per_row * self.n_tax
        anym = ((r >> 8) & 0x7FFFFFFFFFFF) % self.n_keys
        cls = (r & 0xFF) % 100
        ids = torch.where(cls < 52, same, torch.where(cls < 70, anym, 10**12 + ((r >> 8) & 0x7FFFFFFFFFFF)))
        ids = torch.clamp(ids, min=0)
        d, o, tot = self.write_peptides(ids.contiguous(), 0, ids.numel())
        foffs = o[::self.PEPS].contiguous()                     # a fragment = PEPS peptides
        return d, foffs, tot

    # -- host twins (numpy; for the checker) -----------------------------------------------------------------------------------------
    def piece_ids_np(self, b, n_pairs, pairs):
        """Peptide numbers of pairs `pairs` (indices inside batch b): uint64 array (len(pairs), F * PEPS)."""
        per = self.F * self.PEPS
        pairs = np.asarray(pairs, dtype=np.uint64)
        e = (np.uint64(b * n_pairs) + pairs)[:, None] * np.uint64(per) + np.arange(per, dtype=np.uint64)[None, :]
        r = _splitmix_np(e ^ np.uint64(0x5DEECE66D))
        src = (_splitmix_np((e // np.uint64(per)) ^ np.uint64(0x2545F491)) & np.uint64(0x7FFFFFFFFFFF)) % np.uint64(self.n_tax)
        per_row = np.uint64(max(1, self.n_keys // self.n_tax))
        r8 = r >> np.uint64(8)
        same = src + (r8 & np.uint64(0x7FFFFFFF)) % per_row * np.uint64(self.n_tax)
        anym = (r8 & np.uint64(0x7FFFFFFFFFFF)) % np.uint64(self.n_keys)
        cls = (r & np.uint64(0xFF)) % np.uint64(100)
        return np.where(cls < 52, same, np.where(cls < 70, anym, np.uint64(10**12) + (r8 & np.uint64(0x7FFFFFFFFFFF))))

    def lengths_np(self, ids):
        ids = np.asarray(ids, dtype=np.uint64)
        return (np.uint64(14) + (_splitmix_np(ids ^ np.uint64(self.seed)) & np.uint64(0x7FFFFFFF)) % np.uint64(32)).astype(np.int64)

    def peptides_np(self, ids):
        """The residues of peptides `ids` as a list of bytes objects (k_synth_tryptic restated)."""
        ids = np.asarray(ids, dtype=np.uint64).reshape(-1)
        lens = self.lengths_np(ids)
        alphabet = np.frombuffer(b"ACDEFGHILMNPQSTVWY", np.uint8)
        out = np.zeros((len(ids), 48), np.uint8)
        with np.errstate(over="ignore"):
            h = _splitmix_np(np.uint64(self.seed) ^ (ids * np.uint64(0x9E3779B97F4A7C15)))
            for j in range(45):
                if (j & 7) == 0 and j:
                    h = _splitmix_np(h + np.uint64(j))
                d = ((h >> np.uint64(8 * (j & 7))) & np.uint64(0xFF)) % np.uint64(18)
                if j == 0:
                    d = np.where(d == 11, np.uint64(12), d)
                out[:, j] = alphabet[d.astype(np.int64)]
                last = lens - 1 == j                           # the last letter: K or R off the hash state at that point
                if last.any():
                    out[last, j] = np.where((h[last] >> np.uint64(60)) & np.uint64(1), ord("K"), ord("R"))
        # (the device rehashes for j = len at a multiple of 8 only inside its loop, i.e. for j < len: same as above)
        return [out[i, :lens[i]].tobytes() for i in range(len(ids))]

    def values_np(self, ids):
        """Index value of peptide numbers `ids`: the taxon of row (i mod n_tax) for members, 0 for strangers."""
        ids = np.asarray(ids, dtype=np.uint64)
        member = ids < np.uint64(self.n_keys)
        v = self.w.taxon_of_row[(ids % np.uint64(self.n_tax)).astype(np.int64)]
        return np.where(member, v, 0).astype(np.uint32)
