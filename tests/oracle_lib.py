"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
CLI_PATH = os.path.join(ORACLE_DIR, "_build", "umgap_oracle")

NONE = np.uint64(0xFFFFFFFFFFFFFFFF)
ST_OK, ST_EMPTY, ST_UNKNOWN, ST_PANIC = 0, 1, 2, 3
ERRKINDS = ["Msg", "Io", "ParseInt", "ParseFloat", "UnknownRank", "UnknownTaxon", "EmptyInput",
            "InvalidInvocation", "ParseMethod", "ParseStrategy", "Index"]
METHODS = {"tree": 0, "rmq": 1}
STRATEGIES = {"lca*": 0, "hybrid": 1, "mrtl": 2}

_lib = None


def build():
    """(Re)build the oracle with its Makefile if sources are newer than the outputs."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("capi.cpp", "cli.cpp", "umgap_oracle.hpp", "synth_oracle.hpp")]
    outs = [LIB_PATH, CLI_PATH]
    need = not all(os.path.exists(o) for o in outs)
    if not need:
        newest = max(os.path.getmtime(s) for s in srcs)
        need = any(os.path.getmtime(o) < newest for o in outs)
    if need:
        subprocess.run(["make", "-C", ORACLE_DIR, "-j2"], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_taxonomy_from_tsv.restype = C.c_void_p
        L.orc_taxonomy_from_arrays.restype = C.c_void_p
        L.orc_aggregator_new.restype = C.c_void_p
        L.orc_rmq_new.restype = C.c_void_p
        L.orc_sorted_index_new.restype = C.c_void_p
        L.orc_taxonomy_root.restype = C.c_uint64
        L.orc_taxonomy_len.restype = C.c_size_t
        L.orc_taxonomy_count.restype = C.c_size_t
        L.orc_euler.restype = C.c_size_t
        L.orc_rmq_block_min.restype = C.c_size_t
        L.orc_rmq_sparse.restype = C.c_size_t
        L.orc_rmq_query.restype = C.c_uint64
        L.orc_taxa2agg_check.restype = C.c_size_t
        L.orc_seedextend_record.restype = C.c_size_t
        L.orc_rank_score.restype = C.c_int64
        L.orc_taxon_score.restype = C.c_int64
        L.orc_rank_name.restype = C.c_char_p
        L.orc_synth_key_of_member.restype = C.c_uint64
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class SynthIndexParams(C.Structure):
    _fields_ = [("n_keys", C.c_uint64), ("n_tax", C.c_uint64), ("mul_a", C.c_uint64), ("add_b", C.c_uint64)]


def parse_taxon(line: bytes):
    """Returns (id, name, rank, parent, valid) or the ErrKind name."""
    L = lib()
    i, par = C.c_uint64(), C.c_uint64()
    rank, valid = C.c_int(), C.c_int()
    name = C.create_string_buffer(256)
    rc = L.orc_parse_taxon(line, C.c_size_t(len(line)), C.byref(i), name, C.c_size_t(256), C.byref(rank),
                           C.byref(par), C.byref(valid))
    if rc >= 0:
        return ERRKINDS[rc]
    return (i.value, name.value.decode(), rank.value, par.value, bool(valid.value))


def parse_usize(s: bytes):
    out = C.c_uint64()
    rc = lib().orc_parse_usize(s, C.c_size_t(len(s)), C.byref(out))
    return out.value if rc < 0 else ERRKINDS[rc]


def parse_f32(s: bytes):
    out = C.c_float()
    rc = lib().orc_parse_f32(s, C.c_size_t(len(s)), C.byref(out))
    return out.value if rc < 0 else ERRKINDS[rc]


class OracleError(Exception):
    pass


class OraclePanic(Exception):
    pass


class Taxonomy:
    def __init__(self, handle):
        self.h = C.c_void_p(handle)

    @classmethod
    def from_tsv(cls, path):
        err = C.create_string_buffer(512)
        pan = C.c_int()
        h = lib().orc_taxonomy_from_tsv(path.encode(), err, C.c_size_t(512), C.byref(pan))
        if not h:
            raise (OraclePanic if pan.value else OracleError)(err.value.decode())
        return cls(h)

    @classmethod
    def from_arrays(cls, ids, parents, valid=None, rank=None):
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        parents = np.ascontiguousarray(parents, dtype=np.uint64)
        n = len(ids)
        valid = np.ones(n, np.uint8) if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
        rank = np.zeros(n, np.uint8) if rank is None else np.ascontiguousarray(rank, dtype=np.uint8)
        err = C.create_string_buffer(512)
        pan = C.c_int()
        h = lib().orc_taxonomy_from_arrays(_p(ids, C.c_uint64), _p(parents, C.c_uint64), _p(valid, C.c_uint8),
                                           _p(rank, C.c_uint8), C.c_size_t(n), err, C.c_size_t(512), C.byref(pan))
        if not h:
            raise (OraclePanic if pan.value else OracleError)(err.value.decode())
        return cls(h)

    def __del__(self):
        try:
            lib().orc_taxonomy_free(self.h)
        except Exception:
            pass

    @property
    def root(self):
        return lib().orc_taxonomy_root(self.h)

    def __len__(self):
        return lib().orc_taxonomy_len(self.h)

    def rows(self):
        n = lib().orc_taxonomy_count(self.h)
        ids, par = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        valid, rank = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        lib().orc_taxonomy_rows(self.h, _p(ids, C.c_uint64), _p(par, C.c_uint64), _p(valid, C.c_uint8), _p(rank, C.c_uint8))
        return ids, par, valid, rank

    def ancestry(self):
        out = np.zeros(len(self), np.uint64)
        lib().orc_ancestry(self.h, _p(out, C.c_uint64))
        return out

    def snapping(self, ranked=False):
        out = np.zeros(len(self), np.uint64)
        rc = lib().orc_snapping(self.h, int(ranked), _p(out, C.c_uint64))
        if rc:
            raise OraclePanic("snapping")
        return out

    def euler(self):
        cap = 2 * lib().orc_taxonomy_count(self.h) + 2
        t, d = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        n = lib().orc_euler(self.h, _p(t, C.c_uint64), _p(d, C.c_uint64), C.c_size_t(cap))
        return list(zip(t[:n].tolist(), d[:n].tolist()))

    def lineage(self, index):
        out = np.zeros(32, np.uint64)
        bad = C.c_uint64()
        rc = lib().orc_lineage(self.h, C.c_uint64(index), _p(out, C.c_uint64), C.byref(bad))
        if rc:
            raise OracleError("Unknown Taxon ID: %d" % bad.value)
        return [None if v == NONE else int(v) for v in out]

    def score(self, index, with_unknown=False):
        s = lib().orc_taxon_score(self.h, C.c_uint64(index), int(with_unknown))
        return None if s < 0 else s


class AggResult:
    def __init__(self, ok, unknown, empty_input, may_panic):
        self.ok, self.unknown, self.empty_input, self.may_panic = ok, unknown, empty_input, may_panic

    def __repr__(self):
        return "AggResult(ok=%r, unknown=%r, empty=%r, panic=%r)" % (self.ok, self.unknown, self.empty_input, self.may_panic)


class Aggregator:
    def __init__(self, taxonomy, method="tree", strategy="hybrid", factor=0.25):
        self.tax = taxonomy
        err = C.create_string_buffer(512)
        h = lib().orc_aggregator_new(taxonomy.h, METHODS[method], STRATEGIES[strategy], C.c_float(factor), err, C.c_size_t(512))
        if not h:
            raise OracleError(err.value.decode())
        self.h = C.c_void_p(h)

    def __del__(self):
        try:
            lib().orc_aggregator_free(self.h)
        except Exception:
            pass

    def aggregate(self, taxa, scores=None):
        """counting_aggregate (scores None) or aggregate over (taxon, score) pairs."""
        taxa = np.ascontiguousarray(taxa, dtype=np.uint64)
        sc = None if scores is None else np.ascontiguousarray(scores, dtype=np.float32)
        cap = max(16, 4 * len(taxa) + 16)
        ok, unk = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        n_ok, n_unk, flags = C.c_size_t(), C.c_size_t(), C.c_int()
        lib().orc_aggregate(self.h, _p(taxa, C.c_uint64), _p(sc, C.c_float), C.c_size_t(len(taxa)), _p(ok, C.c_uint64),
                            C.byref(n_ok), C.c_size_t(cap), _p(unk, C.c_uint64), C.byref(n_unk), C.c_size_t(cap), C.byref(flags))
        return AggResult(set(ok[:n_ok.value].tolist()), set(unk[:n_unk.value].tolist()), bool(flags.value & 1), bool(flags.value & 2))

    def lca2(self, a, b):
        out = C.c_uint64()
        rc = lib().orc_lca2(self.h, C.c_uint64(a), C.c_uint64(b), C.byref(out))
        assert rc >= 0, "lca2 needs method=rmq strategy=lca*"
        if rc:
            raise OracleError("Unknown Taxon ID: %d" % out.value)
        return out.value

    def taxa2agg_check(self, taxa, offsets, got_taxon=None, got_status=None, scores=None, ranked=False, lower_bound=0.0,
                       nthreads=8, cap_bad=16):
        """Run the taxa2agg record loop over CSR (taxa, offsets); returns (n_bad, bad_idx, want_taxon, want_status)."""
        taxa = np.ascontiguousarray(taxa, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        sc = None if scores is None else np.ascontiguousarray(scores, dtype=np.float32)
        gt = None if got_taxon is None else np.ascontiguousarray(got_taxon, dtype=np.uint32)
        gs = None if got_status is None else np.ascontiguousarray(got_status, dtype=np.uint32)
        bad = np.zeros(cap_bad, np.uint64)
        wt, ws = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        nbad = lib().orc_taxa2agg_check(self.h, int(ranked), C.c_float(lower_bound), _p(taxa, C.c_uint32), _p(sc, C.c_float),
                                        _p(offsets, C.c_uint64), C.c_size_t(n), _p(gt, C.c_uint32), _p(gs, C.c_uint32),
                                        _p(bad, C.c_uint64), C.c_size_t(cap_bad), _p(wt, C.c_uint32), _p(ws, C.c_uint32),
                                        int(nthreads))
        return nbad, bad[:min(nbad, cap_bad)], wt, ws


class RMQ:
    def __init__(self, array):
        self.a = np.ascontiguousarray(array, dtype=np.uint64)
        self.h = C.c_void_p(lib().orc_rmq_new(_p(self.a, C.c_uint64), C.c_size_t(len(self.a))))

    def __del__(self):
        try:
            lib().orc_rmq_free(self.h)
        except Exception:
            pass

    def query(self, s, e):
        return lib().orc_rmq_query(self.h, C.c_uint64(s), C.c_uint64(e))

    @staticmethod
    def block_min(array):
        a = np.ascontiguousarray(array, dtype=np.uint64)
        out = np.zeros(len(a) // 64 + 2, np.uint64)
        n = lib().orc_rmq_block_min(_p(a, C.c_uint64), C.c_size_t(len(a)), _p(out, C.c_uint64))
        return out[:n].tolist()

    @staticmethod
    def sparse(array, block_min):
        a = np.ascontiguousarray(array, dtype=np.uint64)
        b = np.ascontiguousarray(block_min, dtype=np.uint64)
        out = np.zeros(64 * (len(b) + 1), np.uint64)
        lens = np.zeros(64, np.uint64)
        nl = lib().orc_rmq_sparse(_p(a, C.c_uint64), C.c_size_t(len(a)), _p(b, C.c_uint64), C.c_size_t(len(b)),
                                  _p(out, C.c_uint64), _p(lens, C.c_uint64), C.c_size_t(64))
        res, k = [], 0
        for l in range(nl):
            res.append(out[k:k + int(lens[l])].tolist())
            k += int(lens[l])
        return res

    @staticmethod
    def labels(array):
        a = np.ascontiguousarray(array, dtype=np.uint64)
        out = np.zeros(len(a), np.uint64)
        lib().orc_rmq_labels(_p(a, C.c_uint64), C.c_size_t(len(a)), _p(out, C.c_uint64))
        return out


def seedextend_record(taxa, min_seed_size=2, max_gap_size=0):
    """Kept taxa of one record, or None where the reference panics (src/commands/seedextend.rs:169-173)."""
    taxa = np.ascontiguousarray(taxa, dtype=np.uint32)
    out = np.zeros(len(taxa) + 1, np.uint32)
    n = lib().orc_seedextend_record(_p(taxa, C.c_uint32), C.c_size_t(len(taxa)), C.c_size_t(min_seed_size),
                                    C.c_size_t(max_gap_size), _p(out, C.c_uint32))
    if n == C.c_size_t(-1).value:
        return None
    return out[:n]


class RankedList:
    """TaxonList::new_with_unknown(taxa, true) of a Taxonomy: what `seedextend -r` scores against."""

    def __init__(self, taxonomy):
        self.tax = taxonomy
        lib().orc_ranked_list_new.restype = C.c_void_p
        lib().orc_seedextend_record_ranked.restype = C.c_size_t
        self.h = C.c_void_p(lib().orc_ranked_list_new(taxonomy.h))

    def __del__(self):
        try:
            lib().orc_ranked_list_free(self.h)
        except Exception:
            pass

    def seedextend_record(self, taxa, min_seed_size=2, max_gap_size=0, penalty=5, overflow_checks=False):
        """Kept taxa of one record in ranked mode, or None where the reference panics (src/commands/seedextend.rs:150-173).
        overflow_checks: a debug build (`e - s` of an inverted seed panics); default: a release build (it wraps)."""
        taxa = np.ascontiguousarray(taxa, dtype=np.uint32)
        out = np.zeros(len(taxa) + 1, np.uint32)
        n = lib().orc_seedextend_record_ranked(self.h, _p(taxa, C.c_uint32), C.c_size_t(len(taxa)), C.c_size_t(min_seed_size),
                                               C.c_size_t(max_gap_size), C.c_size_t(penalty), C.c_int(int(overflow_checks)), _p(out, C.c_uint32))
        if n == C.c_size_t(-1).value:
            return None
        return out[:n]


def synth_lookup9(params: SynthIndexParams, taxon_of_row, keys45, miss_value=0):
    tor = np.ascontiguousarray(taxon_of_row, dtype=np.uint32)
    keys = np.ascontiguousarray(keys45, dtype=np.uint64)
    out = np.zeros(len(keys), np.uint32)
    lib().orc_synth_lookup9(C.byref(params), _p(tor, C.c_uint32), _p(keys, C.c_uint64), C.c_size_t(len(keys)),
                            C.c_uint32(miss_value), _p(out, C.c_uint32))
    return out


def synth_key_of_member(params: SynthIndexParams, member: int) -> int:
    return lib().orc_synth_key_of_member(C.byref(params), C.c_uint64(member))


class SortedIndex:
    """CPU-baseline index: sorted u64 keys + binary search (dependent accesses), threaded."""

    def __init__(self, keys, vals):
        order = np.argsort(keys, kind="stable")
        self.keys = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64)[order])
        self.vals = np.ascontiguousarray(np.asarray(vals, dtype=np.uint32)[order])
        self.h = C.c_void_p(lib().orc_sorted_index_new(_p(self.keys, C.c_uint64), _p(self.vals, C.c_uint32), C.c_size_t(len(self.keys))))

    def __del__(self):
        try:
            lib().orc_sorted_index_free(self.h)
        except Exception:
            pass

    def lookup9(self, keys, miss_value=0, nthreads=1):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(len(keys), np.uint32)
        lib().orc_sorted_lookup9(self.h, _p(keys, C.c_uint64), C.c_size_t(len(keys)), C.c_uint32(miss_value), _p(out, C.c_uint32), int(nthreads))
        return out


def run_cli(args, stdin: bytes):
    """Run the oracle CLI; returns (returncode, stdout bytes, stderr bytes)."""
    build()
    p = subprocess.run([CLI_PATH] + list(args), input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr
