"""ctypes binding of libumgap_hip.so (the C ABI of include/umgap_hip.h).

The library is built in tree by umgap_amd/csrc/Makefile (see __graft_entry__.build()).  There is
no CPU fallback: if the shared object is missing this module raises, and every compute entry
point of the library itself fails with UMGAP_ERR_DEVICE when no GPU is usable.
"""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libumgap_hip.so")
HEADER_PATH = os.path.join(ROOT, "include", "umgap_hip.h")

OK = 0
ERR_INVALID, ERR_IO, ERR_FORMAT, ERR_DEVICE, ERR_NOMEM, ERR_BUILD, ERR_PANIC = -1, -2, -3, -4, -5, -6, -7
READ_OK, READ_EMPTY, READ_UNKNOWN_TAXON, READ_PANIC = 0, 1, 2, 3
METHOD_TREE, METHOD_RMQ = 0, 1
STRATEGY_LCA_STAR, STRATEGY_HYBRID, STRATEGY_MRTL = 0, 1, 2
NONE = 0xFFFFFFFF

_lib = None


class UmgapError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("umgap error %d: %s" % (code, msg))
        self.code = code
        self.message = msg


def declared_symbols():
    """Every function name include/umgap_hip.h declares."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(umgap_[a-z0-9_]+)\s*\(", text)))


def load():
    """Loads the shared library (importing torch first when it is present, so that both use the
    same libamdhip64: torch bundles one with the same SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C umgap_amd/csrc` (or __graft_entry__.build()). "
            "umgap_amd has no CPU fallback." % LIB_PATH)
    try:
        import torch  # noqa: F401  (plumbing only: shares its HIP runtime with the library)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, sz, i32, f32, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t, C.c_int, C.c_float, C.c_double
    pp = C.POINTER(C.c_void_p)
    sig = {
        "umgap_last_error": (C.c_char_p, []),
        "umgap_version": (C.c_char_p, []),
        "umgap_device_count": (i32, []),
        "umgap_pack9": (i32, [C.c_char_p, C.POINTER(u64)]),
        "umgap_unpack9": (None, [u64, C.c_char_p]),
        "umgap_index_build9": (i32, [vp, vp, sz, dbl, i32, pp]),
        "umgap_index_build9_dev": (i32, [vp, vp, sz, dbl, i32, pp]),
        "umgap_index_save": (i32, [vp, C.c_char_p]),
        "umgap_index_open": (i32, [C.c_char_p, u32, pp]),
        "umgap_index_open_shard": (i32, [C.c_char_p, u32, u32, u32, pp]),
        "umgap_index_set_shard": (i32, [vp, u32, u32]),
        "umgap_index_owner9_dev": (i32, [vp, vp, sz, u32, vp, vp]),
        "umgap_index_partition9_dev": (i32, [vp, vp, sz, u32, u64, vp, vp, vp, vp, vp]),
        "umgap_index_free": (None, [vp]),
        "umgap_index_num_keys": (u64, [vp]),
        "umgap_index_num_buckets": (u64, [vp]),
        "umgap_index_table_bytes": (u64, [vp]),
        "umgap_index_device": (i32, [vp]),
        "umgap_index_lookup9": (i32, [vp, vp, sz, u32, vp]),
        "umgap_index_lookup_records": (i32, [vp, vp, vp, sz, u32, vp]),
        "umgap_index_walk_info": (i32, [vp, vp, vp, vp]),
        "umgap_index_lookup9_dev": (i32, [vp, vp, sz, u32, vp, vp]),
        "umgap_index_lookup9_lane_dev": (i32, [vp, vp, sz, u32, vp, vp]),
        "umgap_index_lookup_prot9_dev": (i32, [vp, vp, vp, vp, sz, u32, vp, vp]),
        "umgap_index_kind": (i32, [vp]),
        "umgap_index_var_layout": (i32, [vp]),
        "umgap_index_build_var": (i32, [vp, vp, vp, sz, dbl, i32, pp]),
        "umgap_index_build_var_dev": (i32, [vp, vp, vp, sz, u64, dbl, i32, pp]),
        "umgap_index_lookup_var": (i32, [vp, vp, vp, sz, u32, vp]),
        "umgap_index_lookup_var_dev": (i32, [vp, vp, vp, sz, u32, vp, vp]),
        "umgap_index_dump9": (i32, [vp, vp, vp, sz, vp]),
        "umgap_index_dump_var": (i32, [vp, vp, u64, vp, vp, vp, sz, vp, vp]),
        "umgap_taxonomy_create": (i32, [vp, vp, vp, vp, sz, i32, pp]),
        "umgap_taxonomy_free": (None, [vp]),
        "umgap_taxonomy_root": (u32, [vp]),
        "umgap_taxonomy_len": (u32, [vp]),
        "umgap_taxonomy_reachable": (u32, [vp]),
        "umgap_taxonomy_snapping": (i32, [vp, i32, vp]),
        "umgap_taxonomy_preorder": (i32, [vp, vp, vp, vp]),
        "umgap_aggregate": (i32, [vp, i32, i32, f32, f32, i32, vp, vp, vp, sz, vp, vp]),
        "umgap_aggregate_dev": (i32, [vp, i32, i32, f32, f32, i32, vp, vp, vp, sz, vp, vp, vp]),
        "umgap_aggregate_strided_dev": (i32, [vp, i32, i32, f32, f32, i32, vp, u32, sz, vp, vp, vp]),
        "umgap_classify9_dev": (i32, [vp, vp, i32, f32, i32, vp, sz, u32, u32, u32, u32, vp, vp, vp]),
        "umgap_classify9_strategy_dev": (i32, [vp, vp, i32, i32, f32, f32, i32, vp, sz, u32, u32, u32, u32, vp, vp, vp]),
        "umgap_classify_prot9_dev": (i32, [vp, vp, i32, f32, i32, vp, vp, sz, u32, i32, u32, u32, vp, vp, vp]),
        "umgap_classify_prot9_strategy_dev": (i32, [vp, vp, i32, i32, f32, f32, i32, vp, vp, sz, u32, i32, u32, u32, vp, vp, vp]),
        "umgap_index_builder_new": (i32, [u64, dbl, i32, pp]),
        "umgap_index_builder_add": (i32, [vp, vp, vp, sz]),
        "umgap_index_builder_finish": (i32, [vp, pp]),
        "umgap_index_builder_free": (None, [vp]),
        "umgap_index_bind_taxonomy": (i32, [vp, vp]),
        "umgap_index_unbind_taxonomy": (i32, [vp, vp]),
        "umgap_index_is_bound": (i32, [vp]),
        "umgap_classify_records": (i32, [vp, vp, i32, i32, f32, f32, i32, vp, vp, sz, vp, sz, i32, u32, u32, vp, vp]),
        "umgap_classify_records_sg": (i32, [vp, vp, i32, i32, f32, f32, i32, vp, vp, sz, vp, sz, vp, sz, i32, u32, u32, vp, vp]),
        "umgap_records_last_route": (i32, []),
        "umgap_comm_unique_id": (i32, [vp]),
        "umgap_comm_init_rank": (i32, [vp, i32, i32, i32, pp]),
        "umgap_comm_init_local": (i32, [i32, i32, pp]),
        "umgap_comm_init_ipc": (i32, [C.c_char_p, i32, i32, i32, pp]),
        "umgap_comm_world": (i32, [vp]),
        "umgap_synth_tryptic_dev": (i32, [vp, u64, sz, u64, vp, vp, i32, vp]),
        "umgap_digest_tryp_dev": (i32, [vp, vp, sz, vp, sz, C.POINTER(sz), i32, vp]),
        "umgap_classify_tryp_dev": (i32, [vp, vp, i32, i32, f32, f32, i32, vp, vp, sz, u32, u32, u32, vp, vp, C.POINTER(u64), vp]),
        "umgap_comm_rank": (i32, [vp]),
        "umgap_comm_selftest": (i32, [vp, sz]),
        "umgap_comm_free": (None, [vp]),
        "umgap_sharded_create": (i32, [vp, vp, vp, u64, u32, dbl, pp]),
        "umgap_sharded_free": (None, [vp]),
        "umgap_sharded_value_bytes": (u32, [vp]),
        "umgap_sharded_region_slots": (u64, [vp]),
        "umgap_sharded_pieces": (u32, [vp]),
        "umgap_sharded_transport": (i32, [vp]),
        "umgap_sharded_lookup9_dev": (i32, [vp, vp, sz, u32, vp, vp]),
        "umgap_sharded_lookup9": (i32, [vp, vp, sz, u32, vp]),
        "umgap_sharded_classify9_dev": (i32, [vp, i32, i32, f32, f32, i32, vp, sz, u32, u32, u32, u32, vp, vp, vp]),
        "umgap_sharded_check": (i32, [vp]),
        "umgap_sharded_probe_ms": (i32, [vp, C.POINTER(C.c_float)]),
        "umgap_host_alloc": (i32, [sz, pp]),
        "umgap_host_free": (None, [vp]),
        "umgap_index_build_windows9_dev": (i32, [vp, vp, vp, sz, u32, u64, dbl, i32, pp]),
        "umgap_index_build_walk9_dev": (i32, [vp, vp, vp, sz, u32, u64, dbl, i32, pp]),
        "umgap_walk_builder_new": (i32, [vp, dbl, pp]),
        "umgap_walk_builder_add_dev": (i32, [vp, vp, vp, sz, u32]),
        "umgap_walk_builder_add": (i32, [vp, vp, vp, sz, u32]),
        "umgap_walk_builder_finish": (i32, [vp, pp]),
        "umgap_walk_builder_new_deferred": (i32, [u64, dbl, i32, pp]),
        "umgap_walk_builder_set_values_dev": (i32, [vp, vp, vp, sz]),
        "umgap_walk_builder_set_values": (i32, [vp, vp, vp, sz]),
        "umgap_walk_builder_free": (None, [vp]),
        "umgap_synth_prot_reads_dev": (i32, [vp, u64, u32, u64, u64, sz, u32, u32, vp, i32, vp]),
        "umgap_seedextend_dev": (i32, [vp, sz, u32, u32, vp, vp, i32, vp]),
        "umgap_seedextend_ranked_dev": (i32, [vp, vp, sz, u32, u32, u32, vp, vp, vp]),
        "umgap_seedextend_strided_dev": (i32, [u32, sz, u32, u32, vp, vp, i32, vp]),
        "umgap_bench_gather_dev": (i32, [vp, sz, u32, vp, vp]),
        "umgap_synth_index_build": (i32, [u64, vp, u64, u64, u64, dbl, i32, pp]),
        "umgap_synth_index_build_shard": (i32, [u64, vp, u64, u64, u64, dbl, i32, u32, u32, pp]),
        "umgap_synth_peptides_dev": (i32, [u64, sz, u64, vp, vp, i32, vp]),
        "umgap_synth_reads9_dev": (i32, [vp, vp, vp, u64, u64, u64, u64, u64, u64, u64, u32, vp, vp]),
        "umgap_synth_reads9_mix_dev": (i32, [vp, vp, vp, u64, u64, u64, u64, u64, u64, u64, u32, u32, u32, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the build is stale
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise UmgapError(rc, load().umgap_last_error().decode(errors="replace"))
