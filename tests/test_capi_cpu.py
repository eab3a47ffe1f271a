"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/umgap_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import umgap_amd
from umgap_amd import _lib


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(_lib.HERE, "csrc"), "-j4", "lib"], check=True, stdout=subprocess.DEVNULL)
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    names = umgap_amd.declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libumgap_hip.so does not export %s" % n


def test_version_is_the_reference_version(lib):
    assert lib.umgap_version() == b"1.1.1"  # scripts/umgap-analyse.sh:107 probes `umgap -V`


def test_pack9_convention(lib):
    k = C.c_uint64()
    assert lib.umgap_pack9(b"AAAAAAAAA", C.byref(k)) == 1 and k.value == 0
    assert lib.umgap_pack9(b"AAAAAAAAB", C.byref(k)) == 1 and k.value == 1
    assert lib.umgap_pack9(b"ZAAAAAAAA", C.byref(k)) == 1 and k.value == 25 << 40
    assert lib.umgap_pack9(b"AAAA*AAAA", C.byref(k)) == 0
    assert lib.umgap_pack9(b"aAAAAAAAA", C.byref(k)) == 0
    buf = C.create_string_buffer(9)
    lib.umgap_pack9(b"ENFVYLAKW", C.byref(k))
    lib.umgap_unpack9(k, buf)
    assert buf.raw[:9] == b"ENFVYLAKW"


def _has_gpu(lib):
    return lib.umgap_device_count() > 0


def test_no_cpu_fallback(lib):
    if _has_gpu(lib):
        pytest.skip("a GPU is visible")
    keys = np.arange(4, dtype=np.uint64)
    vals = np.arange(4, dtype=np.uint32)
    h = C.c_void_p()
    rc = lib.umgap_index_build9(keys.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), 4, 0.0, 0, C.byref(h))
    assert rc == _lib.ERR_DEVICE and b"no CPU fallback" in lib.umgap_last_error()
    ids = np.array([1, 2], np.uint32)
    par = np.array([1, 1], np.uint32)
    ones = np.ones(2, np.uint8)
    rc = lib.umgap_taxonomy_create(ids.ctypes.data_as(C.c_void_p), par.ctypes.data_as(C.c_void_p), ones.ctypes.data_as(C.c_void_p),
                                   ones.ctypes.data_as(C.c_void_p), 2, 0, C.byref(h))
    assert rc == _lib.ERR_DEVICE


def test_round2_entry_points_fail_loudly_without_gpu(lib):
    """Pinned buffers and communicators need the device too: nothing of the sharded path has a host-side stand-in."""
    if _has_gpu(lib):
        pytest.skip("a GPU is visible")
    p = C.c_void_p()
    assert lib.umgap_host_alloc(4096, C.byref(p)) == _lib.ERR_DEVICE and not p.value
    ranks = (C.c_void_p * 2)()
    assert lib.umgap_comm_init_local(2, 0, ranks) == _lib.ERR_DEVICE
    assert lib.umgap_comm_init_local(0, 0, ranks) == _lib.ERR_INVALID
    assert lib.umgap_comm_world(None) == 0 and lib.umgap_comm_rank(None) == -1
    h = C.c_void_p()
    assert lib.umgap_sharded_create(None, None, None, 1024, 1, 0.1, C.byref(h)) == _lib.ERR_INVALID


def test_taxonomy_validation_mirrors_reference_panics(lib):
    """TaxonTree::new panics (src/taxon.rs:226,239-246) surface as UMGAP_ERR_PANIC before any GPU use."""
    def create(ids, par):
        ids, par = np.array(ids, np.uint32), np.array(par, np.uint32)
        ones = np.ones(len(ids), np.uint8)
        h = C.c_void_p()
        rc = lib.umgap_taxonomy_create(ids.ctypes.data_as(C.c_void_p), par.ctypes.data_as(C.c_void_p), ones.ctypes.data_as(C.c_void_p),
                                       ones.ctypes.data_as(C.c_void_p), len(ids), 0, C.byref(h))
        return rc, lib.umgap_last_error()
    rc, msg = create([1, 2], [1, 2])
    assert rc == _lib.ERR_PANIC and b"More than one root!" in msg
    rc, msg = create([1, 2], [2, 1])
    assert rc == _lib.ERR_PANIC and b"There's no root!" in msg
    rc, msg = create([], [])
    assert rc == _lib.ERR_PANIC
    rc, msg = create([1, 2, 3, 2], [1, 1, 1, 3])          # the same id under two parents: no tree (same-parent repeats are accepted)
    assert rc == _lib.ERR_FORMAT and b"duplicate" in msg and b"different parents" in msg
    rc, msg = create([1, 2, 3], [1, 3, 2])
    assert rc == _lib.ERR_FORMAT and b"cycle" in msg


def test_synthetic_taxonomy_is_a_tree():
    from umgap_amd.synth import synthetic_taxonomy
    import oracle_lib as O
    ids, par, valid, rank = synthetic_taxonomy(5000, 7000, max_depth=30)
    assert len(set(ids.tolist())) == 5000 and 1 in ids
    t = O.Taxonomy.from_arrays(ids, par, valid, rank)
    assert t.root == 1
    depth = max(d for _, d in t.euler())
    assert 25 <= depth <= 30
    snap = t.snapping(False)
    assert all(snap[i] != O.NONE for i in ids)  # every taxon reachable from the root


def test_synth_key_spec_matches_oracle():
    from umgap_amd import synth
    import oracle_lib as O
    p = O.SynthIndexParams(1000, 10, synth.MUL_A, synth.ADD_B)
    for i in (0, 1, 999, 123456789, synth.M20_9 - 1):
        assert synth.key_of_member(i) == O.synth_key_of_member(p, i)
    keys = np.array([synth.key_of_member(i) for i in range(0, 2000, 7)], np.uint64)
    tor = np.arange(100, 110, dtype=np.uint32)
    got = O.synth_lookup9(p, tor, keys, miss_value=0)
    want = np.array([tor[i % 10] if i < 1000 else 0 for i in range(0, 2000, 7)], np.uint32)
    assert (got == want).all()
