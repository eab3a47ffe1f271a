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
