"""Host-side mirror of the reference's interface for the pept2lca -> taxa2agg path, over the
C ABI (include/umgap_hip.h).  PyTorch is used only as plumbing (device memory and streams).

  Index      <-> fst::Map               (src/commands/pept2lca.rs:74-79,93)
  Taxonomy   <-> TaxonList + TaxonTree  (src/commands/taxa2agg.rs:104-109)
  Taxonomy.aggregate / Aggregator <-> trait agg::Aggregator + the taxa2agg record loop
                                        (src/agg/mod.rs:12-21, src/commands/taxa2agg.rs:161-180)
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (METHOD_RMQ, METHOD_TREE, NONE, READ_EMPTY, READ_OK, READ_PANIC, READ_UNKNOWN_TAXON,
                   STRATEGY_HYBRID, STRATEGY_LCA_STAR, STRATEGY_MRTL, UmgapError, check)

METHODS = {"tree": METHOD_TREE, "rmq": METHOD_RMQ}          # src/commands/taxa2agg.rs:193
STRATEGIES = {"lca*": STRATEGY_LCA_STAR, "hybrid": STRATEGY_HYBRID, "mrtl": STRATEGY_MRTL}  # :222

RANKS = ["no rank", "superkingdom", "domain", "realm", "kingdom", "subkingdom", "superphylum", "phylum",
         "subphylum", "superclass", "class", "subclass", "infraclass", "superorder", "order", "suborder",
         "infraorder", "parvorder", "superfamily", "family", "subfamily", "tribe", "subtribe", "genus",
         "subgenus", "species group", "species subgroup", "species", "subspecies", "varietas", "forma",
         "strain"]  # src/rank.rs:10-41


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _stream_ptr(stream):
    if stream is None:
        import torch
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def _dev_ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _concat(strings):
    """list of bytes -> (uint8 data, uint64 offsets[n+1])"""
    offs = np.zeros(len(strings) + 1, np.uint64)
    if len(strings):
        offs[1:] = np.cumsum([len(s) for s in strings])
    data = np.frombuffer(b"".join(strings), dtype=np.uint8) if len(strings) else np.zeros(0, np.uint8)
    if data.size == 0:
        data = np.zeros(1, np.uint8)
    return np.ascontiguousarray(data), offs


def pack9(peptide: bytes):
    """45-bit packed key of a 9-mer, or None if a byte is outside 'A'..'Z'."""
    if len(peptide) != 9:
        return None
    k = C.c_uint64()
    return k.value if _lib.load().umgap_pack9(peptide, C.byref(k)) else None


def pack9_array(peptides):
    """Vector form: (keys uint64[n], ok bool[n])."""
    keys = np.zeros(len(peptides), np.uint64)
    ok = np.zeros(len(peptides), bool)
    for i, p in enumerate(peptides):
        k = pack9(p)
        if k is not None:
            keys[i], ok[i] = k, True
    return keys, ok


def unpack9(key: int) -> bytes:
    buf = C.create_string_buffer(9)
    _lib.load().umgap_unpack9(C.c_uint64(key), buf)
    return buf.raw[:9]


class Index:
    """The 9-mer -> taxon table in HBM."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    # -- constructors ----------------------------------------------------------------------
    @classmethod
    def build(cls, keys45, taxa, load_factor=0.0, device=0):
        """From host arrays: the `umgap buildindex` equivalent (src/commands/buildindex.rs:32-48)."""
        keys = np.ascontiguousarray(keys45, dtype=np.uint64)
        vals = np.ascontiguousarray(taxa, dtype=np.uint32)
        if len(keys) != len(vals):
            raise ValueError("keys and taxa differ in length")
        h = C.c_void_p()
        check(_lib.load().umgap_index_build9(_np_ptr(keys), _np_ptr(vals), len(keys), load_factor, device, C.byref(h)))
        return cls(h.value)

    @classmethod
    def build_from_tsv(cls, path, load_factor=0.0, device=0):
        """TSV of `9-mer<TAB>taxon` lines as `umgap buildindex` reads them; keys must be [A-Z]{9}."""
        keys, vals = [], []
        with open(path, "rb") as f:
            for line in f:
                line = line.rstrip(b"\r\n")
                if not line:
                    continue
                k, v = line.split(b"\t")
                key = pack9(k)
                if key is None:
                    raise ValueError("index key %r is not [A-Z]{9}" % k)
                keys.append(key)
                vals.append(int(v))
        return cls.build(np.array(keys, np.uint64), np.array(vals, np.uint32), load_factor, device)

    @classmethod
    def build_dev(cls, d_keys45, d_taxa, load_factor=0.0):
        """From torch CUDA tensors (int64 keys viewed as uint64, int32 taxa viewed as uint32)."""
        h = C.c_void_p()
        check(_lib.load().umgap_index_build9_dev(_dev_ptr(d_keys45), _dev_ptr(d_taxa), d_keys45.numel(), load_factor,
                                                 d_keys45.device.index or 0, C.byref(h)))
        return cls(h.value)

    @classmethod
    def build_walk(cls, values, chunks, load_factor=0.0):
        """The walk-order form (kind 10) of the kind-9 index `values`: order from a proteome given as an iterable of chunks
        (d_bytes uint8, d_offsets int64, both torch CUDA tensors; max_fragment_len), in proteome order (include/umgap_hip.h:
        umgap_walk_builder_*).  Same answers as `values`."""
        L = _lib.load()
        b = C.c_void_p()
        check(L.umgap_walk_builder_new(values._h, load_factor, C.byref(b)))
        try:
            for d_bytes, d_offsets, max_len in chunks:
                check(L.umgap_walk_builder_add_dev(b, _dev_ptr(d_bytes), _dev_ptr(d_offsets), d_offsets.numel() - 1, max_len))
            h = C.c_void_p()
            check(L.umgap_walk_builder_finish(b, C.byref(h)))
            b = None
            return cls(h.value)
        finally:
            if b is not None:
                L.umgap_walk_builder_free(b)

    @classmethod
    def build_walk_deferred(cls, expected_keys, chunks, value_chunks, load_factor=0.0, device=0):
        """A walk-order index (kind 10) built WITHOUT a values table in HBM (umgap_walk_builder_new_deferred): the order from the
        proteome (`chunks` as for build_walk), then the values from `value_chunks`: an iterable of (d_keys45 int64, d_values int32) torch
        CUDA tensors -- the (key, value) pairs of the index in any order.  Same answers as the plain table of those pairs."""
        L = _lib.load()
        b = C.c_void_p()
        check(L.umgap_walk_builder_new_deferred(expected_keys, load_factor, device, C.byref(b)))
        try:
            for d_bytes, d_offsets, max_len in chunks:
                check(L.umgap_walk_builder_add_dev(b, _dev_ptr(d_bytes), _dev_ptr(d_offsets), d_offsets.numel() - 1, max_len))
            for d_keys, d_vals in value_chunks:
                check(L.umgap_walk_builder_set_values_dev(b, _dev_ptr(d_keys), _dev_ptr(d_vals), d_keys.numel()))
            h = C.c_void_p()
            check(L.umgap_walk_builder_finish(b, C.byref(h)))
            b = None
            return cls(h.value)
        finally:
            if b is not None:
                L.umgap_walk_builder_free(b)

    @classmethod
    def open(cls, path, device=0, shard=0, n_shards=1):
        h = C.c_void_p()
        check(_lib.load().umgap_index_open_shard(str(path).encode(), 1 << device, shard, n_shards, C.byref(h)))
        return cls(h.value)

    @classmethod
    def synthetic(cls, n_keys, d_taxon_of_row, mul_a, add_b, load_factor=0.0, shard=0, n_shards=1):
        """Closed-form synthetic index generated on the device (DESIGN.md "Synthetic workload");
        with n_shards > 1 only the bucket range of `shard` is built and held."""
        h = C.c_void_p()
        check(_lib.load().umgap_synth_index_build_shard(n_keys, _dev_ptr(d_taxon_of_row), d_taxon_of_row.numel(), mul_a, add_b,
                                                        load_factor, d_taxon_of_row.device.index or 0, shard, n_shards, C.byref(h)))
        return cls(h.value)

    @classmethod
    def build_var(cls, peptides, taxa, load_factor=0.0, device=0):
        """Variable-length index from a list of byte strings (1..63 bytes each)."""
        data, offs = _concat(peptides)
        vals = np.ascontiguousarray(taxa, dtype=np.uint32)
        h = C.c_void_p()
        check(_lib.load().umgap_index_build_var(_np_ptr(data), _np_ptr(offs), _np_ptr(vals), len(peptides), load_factor, device, C.byref(h)))
        return cls(h.value)

    @property
    def kind(self):
        return _lib.load().umgap_index_kind(self._h)

    @property
    def var_layout(self):
        """1: the keys live in the slots (words over 'A'..'Z'), 0: slots + heap, -1: not a byte-string table."""
        return _lib.load().umgap_index_var_layout(self._h)

    def lookup_var(self, peptides, miss_value=0):
        data, offs = _concat(peptides)
        out = np.empty(len(peptides), np.uint32)
        check(_lib.load().umgap_index_lookup_var(self._h, _np_ptr(data), _np_ptr(offs), len(peptides), miss_value, _np_ptr(out)))
        return out

    def lookup_var_dev(self, d_bytes, d_offsets, d_out, miss_value=0, stream=None):
        check(_lib.load().umgap_index_lookup_var_dev(self._h, _dev_ptr(d_bytes), _dev_ptr(d_offsets), d_offsets.numel() - 1, miss_value,
                                                     _dev_ptr(d_out), _stream_ptr(stream)))
        return d_out

    def dump9(self):
        """(keys45, taxa) of a 9-mer table, in table order (umgap_index_dump9: what printindex walks)."""
        L = _lib.load()
        n = C.c_size_t()
        check(L.umgap_index_dump9(self._h, None, None, 0, C.byref(n)))
        keys = np.zeros(max(1, n.value), np.uint64)
        taxa = np.zeros(max(1, n.value), np.uint32)
        check(L.umgap_index_dump9(self._h, _np_ptr(keys), _np_ptr(taxa), n.value, C.byref(n)))
        return keys[:n.value], taxa[:n.value]

    def dump_var(self):
        """(key bytes, offsets, lengths, taxa) of a byte-string table, in table order (umgap_index_dump_var: what printindex walks)."""
        L = _lib.load()
        n, hb = C.c_size_t(), C.c_uint64()
        check(L.umgap_index_dump_var(self._h, None, 0, None, None, None, 0, C.byref(n), C.byref(hb)))
        cap = n.value
        heap, offs = np.zeros(max(hb.value, 1), np.uint8), np.zeros(max(cap, 1), np.uint64)
        lens, taxa = np.zeros(max(cap, 1), np.uint32), np.zeros(max(cap, 1), np.uint32)
        check(L.umgap_index_dump_var(self._h, _np_ptr(heap), hb.value, _np_ptr(offs), _np_ptr(lens), _np_ptr(taxa), cap, C.byref(n), C.byref(hb)))
        return heap[:hb.value], offs[:cap], lens[:cap], taxa[:cap]

    def save(self, path):
        check(_lib.load().umgap_index_save(self._h, str(path).encode()))

    def bind(self, taxonomy):
        """Rewrite the table values into {preorder, subtree size} codes of `taxonomy` (umgap_index_bind_taxonomy)."""
        check(_lib.load().umgap_index_bind_taxonomy(self._h, taxonomy._h))

    def unbind(self, taxonomy):
        check(_lib.load().umgap_index_unbind_taxonomy(self._h, taxonomy._h))

    @property
    def bound(self):
        return bool(_lib.load().umgap_index_is_bound(self._h))

    def close(self):
        if self._h:
            _lib.load().umgap_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- properties --------------------------------------------------------------------------
    @property
    def num_keys(self):
        return _lib.load().umgap_index_num_keys(self._h)

    @property
    def num_buckets(self):
        return _lib.load().umgap_index_num_buckets(self._h)

    @property
    def table_bytes(self):
        return _lib.load().umgap_index_table_bytes(self._h)

    @property
    def device(self):
        return _lib.load().umgap_index_device(self._h)

    def set_shard(self, shard, n_shards):
        check(_lib.load().umgap_index_set_shard(self._h, shard, n_shards))

    # -- get() -------------------------------------------------------------------------------
    def lookup9(self, keys45, miss_value=0):
        """Host arrays in, host array out (fst::Map::get over a batch; miss -> miss_value)."""
        keys = np.ascontiguousarray(keys45, dtype=np.uint64)
        out = np.empty(len(keys), np.uint32)
        check(_lib.load().umgap_index_lookup9(self._h, _np_ptr(keys), len(keys), miss_value, _np_ptr(out)))
        return out

    def walk_info(self):
        """(shift, entries, exceptions) of a walk-order index (include/umgap_hip.h: umgap_index_walk_info)."""
        sh, n, ex = C.c_uint32(), C.c_uint64(), C.c_uint32()
        check(_lib.load().umgap_index_walk_info(self._h, C.byref(sh), C.byref(n), C.byref(ex)))
        return sh.value, n.value, ex.value

    def lookup_records(self, data, frag_offsets, miss_value=0):
        """The windows of protein records cut and looked up on the device (umgap_index_lookup_records): `data` uint8 residues,
        `frag_offsets` uint64 [n + 1]; returns a uint32 array with one entry per byte position of [frag_offsets[0], frag_offsets[n])
        -- the hit of the 9-mer that starts there, miss_value where none does."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offs = np.ascontiguousarray(frag_offsets, dtype=np.uint64)
        out = np.full(int(offs[-1] - offs[0]) if len(offs) else 0, 0xABABABAB, np.uint32)
        check(_lib.load().umgap_index_lookup_records(self._h, _np_ptr(data), _np_ptr(offs), len(offs) - 1, miss_value, _np_ptr(out)))
        return out

    def lookup9_dev(self, d_keys45, d_out=None, miss_value=0, stream=None, variant="quad"):
        """torch CUDA tensors (int64 keys, int32 out); enqueues on `stream` (current stream if None)."""
        import torch
        if d_out is None:
            d_out = torch.empty(d_keys45.numel(), dtype=torch.int32, device=d_keys45.device)
        fn = _lib.load().umgap_index_lookup9_dev if variant == "quad" else _lib.load().umgap_index_lookup9_lane_dev
        check(fn(self._h, _dev_ptr(d_keys45), d_keys45.numel(), miss_value, _dev_ptr(d_out), _stream_ptr(stream)))
        return d_out

    def lookup_prot9_dev(self, d_bytes, d_offsets, d_out_offsets, d_out, miss_value=0, stream=None):
        check(_lib.load().umgap_index_lookup_prot9_dev(self._h, _dev_ptr(d_bytes), _dev_ptr(d_offsets), _dev_ptr(d_out_offsets),
                                                       d_offsets.numel() - 1, miss_value, _dev_ptr(d_out), _stream_ptr(stream)))
        return d_out

    def owner9_dev(self, d_keys45, n_shards, d_owner=None, stream=None):
        import torch
        if d_owner is None:
            d_owner = torch.empty(d_keys45.numel(), dtype=torch.int32, device=d_keys45.device)
        check(_lib.load().umgap_index_owner9_dev(self._h, _dev_ptr(d_keys45), d_keys45.numel(), n_shards, _dev_ptr(d_owner),
                                                 _stream_ptr(stream)))
        return d_owner

    def bench_gather_dev(self, n, bytes_per_probe, d_out, stream=None):
        check(_lib.load().umgap_bench_gather_dev(self._h, n, bytes_per_probe, _dev_ptr(d_out), _stream_ptr(stream)))


def read_taxa_file(path):
    """taxons.tsv -> (id, parent, valid, rank) arrays in row order (src/taxon.rs:65-128)."""
    ids, parents, valid, rank = [], [], [], []
    rank_idx = {n: i for i, n in enumerate(RANKS)}
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()  # BufRead::lines: the remainder after the final newline is not a line
    for raw in lines:
        if raw.endswith(b"\r"):
            raw = raw[:-1]
        fields = raw.decode("utf-8").rstrip().split("\t")
        if len(fields) != 5:
            raise ValueError("Taxon requires five fields")
        if fields[4] not in ("\x01", "\x00"):
            raise ValueError("Couldn't parse the valid byte")
        ids.append(int(fields[0]))
        parents.append(int(fields[3]))
        rank.append(rank_idx[fields[2]])
        valid.append(1 if fields[4] == "\x01" else 0)
    return (np.array(ids, np.uint32), np.array(parents, np.uint32), np.array(valid, np.uint8), np.array(rank, np.uint8))


class Taxonomy:
    """The taxonomy in its preorder-interval encoding on the GPU (umgap_amd/csrc/taxonomy.hpp)."""

    def __init__(self, ids, parents, valid=None, rank=None, device=0):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        parents = np.ascontiguousarray(parents, dtype=np.uint32)
        n = len(ids)
        valid = np.ones(n, np.uint8) if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
        rank = np.zeros(n, np.uint8) if rank is None else np.ascontiguousarray(rank, dtype=np.uint8)
        h = C.c_void_p()
        check(_lib.load().umgap_taxonomy_create(_np_ptr(ids), _np_ptr(parents), _np_ptr(valid), _np_ptr(rank), n, device, C.byref(h)))
        self._h = h
        self.ids = ids

    @classmethod
    def from_tsv(cls, path, device=0):
        return cls(*read_taxa_file(path), device=device)

    def close(self):
        if self._h:
            _lib.load().umgap_taxonomy_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def root(self):
        return _lib.load().umgap_taxonomy_root(self._h)

    def __len__(self):
        return _lib.load().umgap_taxonomy_len(self._h)

    @property
    def reachable(self):
        return _lib.load().umgap_taxonomy_reachable(self._h)

    def snapping(self, ranked_only=False):
        out = np.empty(len(self), np.uint32)
        check(_lib.load().umgap_taxonomy_snapping(self._h, int(ranked_only), _np_ptr(out)))
        return out

    def preorder(self):
        pre, last, depth = (np.empty(len(self), np.uint32) for _ in range(3))
        check(_lib.load().umgap_taxonomy_preorder(self._h, _np_ptr(pre), _np_ptr(last), _np_ptr(depth)))
        return pre, last, depth

    # -- aggregate ---------------------------------------------------------------------------
    def aggregate(self, taxa, read_offsets, method="tree", strategy="hybrid", factor=0.25, lower_bound=0.0,
                  ranked_only=False, scores=None):
        """taxa2agg over CSR host arrays; returns (out_taxon, out_status)."""
        taxa = np.ascontiguousarray(taxa, dtype=np.uint32)
        offs = np.ascontiguousarray(read_offsets, dtype=np.uint64)
        sc = None if scores is None else np.ascontiguousarray(scores, dtype=np.float32)
        n = len(offs) - 1
        out_t, out_s = np.empty(n, np.uint32), np.empty(n, np.uint32)
        check(_lib.load().umgap_aggregate(self._h, METHODS[method], STRATEGIES[strategy], factor, lower_bound, int(ranked_only),
                                          _np_ptr(taxa), _np_ptr(sc) if sc is not None else None, _np_ptr(offs), n,
                                          _np_ptr(out_t), _np_ptr(out_s)))
        return out_t, out_s

    def aggregate_dev(self, d_taxa, d_offsets, d_out_taxon, d_out_status, method="tree", strategy="hybrid", factor=0.25,
                      lower_bound=0.0, ranked_only=False, d_scores=None, stream=None):
        check(_lib.load().umgap_aggregate_dev(self._h, METHODS[method], STRATEGIES[strategy], factor, lower_bound, int(ranked_only),
                                              _dev_ptr(d_taxa), _dev_ptr(d_scores), _dev_ptr(d_offsets), d_offsets.numel() - 1,
                                              _dev_ptr(d_out_taxon), _dev_ptr(d_out_status), _stream_ptr(stream)))

    def aggregate_strided_dev(self, d_taxa, stride, n_reads, d_out_taxon, d_out_status, method="tree", strategy="hybrid",
                              factor=0.25, lower_bound=0.0, ranked_only=False, stream=None):
        check(_lib.load().umgap_aggregate_strided_dev(self._h, METHODS[method], STRATEGIES[strategy], factor, lower_bound,
                                                      int(ranked_only), _dev_ptr(d_taxa), stride, n_reads, _dev_ptr(d_out_taxon),
                                                      _dev_ptr(d_out_status), _stream_ptr(stream)))


def classify9_dev(index, taxonomy, d_keys45, n_reads, kmers_per_read, d_out_taxon, d_out_status, method="tree", lower_bound=0.0,
                  ranked_only=False, fragments_per_read=0, min_seed_size=2, max_gap_size=0, stream=None, strategy="lca*", factor=0.25):
    """Fused probe -> [seedextend] -> aggregate over packed 9-mers (include/umgap_hip.h: umgap_classify9_dev for LCA*,
    umgap_classify9_strategy_dev for MRTL and the hybrids)."""
    if strategy == "lca*":
        check(_lib.load().umgap_classify9_dev(index._h, taxonomy._h, METHODS[method], lower_bound, int(ranked_only), _dev_ptr(d_keys45), n_reads,
                                              kmers_per_read, fragments_per_read, min_seed_size, max_gap_size, _dev_ptr(d_out_taxon),
                                              _dev_ptr(d_out_status), _stream_ptr(stream)))
    else:
        check(_lib.load().umgap_classify9_strategy_dev(index._h, taxonomy._h, METHODS[method], STRATEGIES[strategy], factor, lower_bound,
                                                       int(ranked_only), _dev_ptr(d_keys45), n_reads, kmers_per_read, fragments_per_read,
                                                       min_seed_size, max_gap_size, _dev_ptr(d_out_taxon), _dev_ptr(d_out_status),
                                                       _stream_ptr(stream)))


def classify_prot9_dev(index, taxonomy, d_bytes, d_frag_offsets, n_reads, fragments_per_read, d_out_taxon, d_out_status, method="tree",
                       lower_bound=0.0, ranked_only=False, seedextend=False, min_seed_size=2, max_gap_size=0, stream=None, strategy="lca*",
                       factor=0.25):
    """Fused surface B: protein fragments -> 9-mer windows (LDS) -> probe -> [seedextend] -> aggregate (any strategy)."""
    check(_lib.load().umgap_classify_prot9_strategy_dev(index._h, taxonomy._h, METHODS[method], STRATEGIES[strategy], factor, lower_bound,
                                                        int(ranked_only), _dev_ptr(d_bytes), _dev_ptr(d_frag_offsets), n_reads, fragments_per_read,
                                                        int(seedextend), min_seed_size, max_gap_size, _dev_ptr(d_out_taxon), _dev_ptr(d_out_status),
                                                        _stream_ptr(stream)))


def digest_tryp_dev(d_bytes, d_frag_offsets, device=0, stream=None):
    """Tryptic digest on the device (default pattern of `prot2tryp2lca`): returns a device int64 tensor of n_pieces + 1 offsets."""
    import torch
    n_frag = d_frag_offsets.numel() - 1
    n = C.c_size_t()
    check(_lib.load().umgap_digest_tryp_dev(_dev_ptr(d_bytes), _dev_ptr(d_frag_offsets), n_frag, None, 0, C.byref(n), device, _stream_ptr(stream)))
    out = torch.empty(n.value + 1, dtype=torch.int64, device=d_frag_offsets.device)
    check(_lib.load().umgap_digest_tryp_dev(_dev_ptr(d_bytes), _dev_ptr(d_frag_offsets), n_frag, _dev_ptr(out), out.numel(), C.byref(n), device,
                                            _stream_ptr(stream)))
    return out


def classify_tryp_dev(index, taxonomy, d_bytes, d_frag_offsets, n_reads, fragments_per_read, d_out_taxon, d_out_status, method="tree",
                      strategy="hybrid", factor=0.25, lower_bound=0.0, ranked_only=False, min_len=5, max_len=50, stream=None):
    """Fused tryptic pipeline: protein fragments -> digest -> variable-length lookup -> aggregate.  Returns the number of pieces."""
    n = C.c_uint64()
    check(_lib.load().umgap_classify_tryp_dev(index._h, taxonomy._h, METHODS[method], STRATEGIES[strategy], factor, lower_bound, int(ranked_only),
                                              _dev_ptr(d_bytes), _dev_ptr(d_frag_offsets), n_reads, fragments_per_read, min_len, max_len,
                                              _dev_ptr(d_out_taxon), _dev_ptr(d_out_status), C.byref(n), _stream_ptr(stream)))
    return n.value


def classify_records(index, taxonomy, fragments, read_frags, method="tree", strategy="hybrid", factor=0.25, lower_bound=0.0,
                     ranked_only=False, seedextend=True, min_seed_size=2, max_gap_size=0):
    """`prot2kmer2lca -o | seedextend | uniq | taxa2agg` for host-side records: `fragments` = list of bytes (one
    per FASTA record, each >= 9 residues), read r = fragments[read_frags[r]:read_frags[r+1]].  Returns
    (out_taxon, out_status) numpy arrays (include/umgap_hip.h: umgap_classify_records)."""
    offs = np.zeros(len(fragments) + 1, np.uint64)
    if len(fragments):
        offs[1:] = np.cumsum([len(f) for f in fragments], dtype=np.uint64)
    data = np.frombuffer(b"".join(fragments) + b"\0", dtype=np.uint8)
    rf = np.ascontiguousarray(read_frags, dtype=np.uint64)
    n = len(rf) - 1
    out_t, out_s = np.empty(n, np.uint32), np.empty(n, np.uint32)
    check(_lib.load().umgap_classify_records(index._h, taxonomy._h, METHODS[method], STRATEGIES[strategy], factor, lower_bound, int(ranked_only),
                                             _np_ptr(data), _np_ptr(offs), len(fragments), _np_ptr(rf), n, int(seedextend), min_seed_size,
                                             max_gap_size, _np_ptr(out_t), _np_ptr(out_s)))
    return out_t, out_s


def classify_records_arrays(index, taxonomy, data, frag_offsets, read_frags, method="tree", strategy="hybrid", factor=0.25, lower_bound=0.0,
                            ranked_only=False, seedextend=True, min_seed_size=2, max_gap_size=0):
    """umgap_classify_records on arrays as a host program holds them: `data` uint8 residues, `frag_offsets` uint64 [n_fragments + 1],
    `read_frags` uint64 [n_reads + 1].  Returns (out_taxon, out_status)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offs = np.ascontiguousarray(frag_offsets, dtype=np.uint64)
    rf = np.ascontiguousarray(read_frags, dtype=np.uint64)
    n = len(rf) - 1
    out_t, out_s = np.empty(n, np.uint32), np.empty(n, np.uint32)
    check(_lib.load().umgap_classify_records(index._h, taxonomy._h, METHODS[method], STRATEGIES[strategy], factor, lower_bound, int(ranked_only),
                                             _np_ptr(data), _np_ptr(offs), len(offs) - 1, _np_ptr(rf), n, int(seedextend), min_seed_size,
                                             max_gap_size, _np_ptr(out_t), _np_ptr(out_s)))
    return out_t, out_s


def classify_records_pieces(index, taxonomy, pieces, frag_offsets, read_frags, method="tree", strategy="hybrid", factor=0.25, lower_bound=0.0,
                            ranked_only=False, seedextend=True, min_seed_size=2, max_gap_size=0):
    """umgap_classify_records_sg: the residues as a list of uint8 arrays laid end to end (numpy arrays, or HostArray.array views for
    pinned pieces); frag_offsets index the concatenation.  Returns (out_taxon, out_status)."""
    import ctypes
    pieces = [np.ascontiguousarray(x, dtype=np.uint8) for x in pieces]
    ptrs = (ctypes.c_void_p * max(1, len(pieces)))(*[x.ctypes.data if x.size else None for x in pieces])
    ends = np.cumsum([x.size for x in pieces], dtype=np.uint64) if pieces else np.zeros(0, np.uint64)
    offs = np.ascontiguousarray(frag_offsets, dtype=np.uint64)
    rf = np.ascontiguousarray(read_frags, dtype=np.uint64)
    n = len(rf) - 1
    out_t, out_s = np.empty(n, np.uint32), np.empty(n, np.uint32)
    check(_lib.load().umgap_classify_records_sg(index._h, taxonomy._h, METHODS[method], STRATEGIES[strategy], factor, lower_bound, int(ranked_only),
                                                ctypes.cast(ptrs, ctypes.c_void_p), _np_ptr(ends), len(pieces), _np_ptr(offs), len(offs) - 1, _np_ptr(rf), n,
                                                int(seedextend), min_seed_size, max_gap_size, _np_ptr(out_t), _np_ptr(out_s)))
    return out_t, out_s


def seedextend_ranked_dev(taxonomy, d_taxa, d_offsets, min_seed_size=2, max_gap_size=0, penalty=5, d_kept=None, stream=None):
    """seedextend -r (only the best-scoring extended seed of a record stays, src/commands/seedextend.rs:150-163), in place."""
    check(_lib.load().umgap_seedextend_ranked_dev(taxonomy._h, _dev_ptr(d_offsets), d_offsets.numel() - 1, min_seed_size, max_gap_size, penalty,
                                                  _dev_ptr(d_taxa), _dev_ptr(d_kept), _stream_ptr(stream)))


def seedextend_dev(d_taxa, d_offsets=None, stride=None, n_records=None, min_seed_size=2, max_gap_size=0, d_kept=None, stream=None):
    """seedextend (unranked) in place on a torch CUDA int32 tensor (src/commands/seedextend.rs:99-148)."""
    L = _lib.load()
    dev = d_taxa.device.index or 0
    if d_offsets is not None:
        check(L.umgap_seedextend_dev(_dev_ptr(d_offsets), d_offsets.numel() - 1, min_seed_size, max_gap_size, _dev_ptr(d_taxa),
                                     _dev_ptr(d_kept), dev, _stream_ptr(stream)))
    else:
        check(L.umgap_seedextend_strided_dev(stride, n_records, min_seed_size, max_gap_size, _dev_ptr(d_taxa), _dev_ptr(d_kept), dev,
                                             _stream_ptr(stream)))
