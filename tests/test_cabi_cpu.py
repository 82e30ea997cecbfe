"""CPU-only: the C-ABI library loads and exports every symbol include/ocrl_hip.h declares, and the
host-side parameter table agrees with the oracle's (reference) inventory.  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import slate_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ocrl_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "ocrl_hip.h")).read()
    names = sorted(set(re.findall(r"\b(ocrl_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.ocrl_abi_version() == 5


def test_config_struct_matches_the_header():
    """the binding's SlateConfig and INTEGRATION.md's Cfg snippet carry every field of ocrl_slate_config, in order, and the library
    reports the same size (a host that copies a struct one field short would hand ocrl_slate_create 4 bytes of garbage)"""
    from ocrl_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "ocrl_hip.h")).read()
    body = re.search(r"typedef struct ocrl_slate_config \{(.*?)\} ocrl_slate_config;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            fields += [f.strip() for f in decl.split(None, 1)[1].split(",")]
    assert [n for n, _ in _lib.SlateConfig._fields_] == fields
    assert L.ocrl_slate_config_size() == ctypes.sizeof(_lib.SlateConfig) == 4 * len(fields)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    snippet = doc[doc.index("class Cfg(ctypes.Structure):"):doc.index("h = ctypes.c_void_p()")]
    assert re.findall(r'"([a-z_]+)"', snippet) == fields


@pytest.mark.parametrize("over", [dict(obs_size=64), dict(obs_size=128, num_slots=6), dict(obs_size=16, vocab_size=256, num_dec_blocks=2),
                                  dict(obs_size=32, num_slots=6, use_bcdec=True)])
def test_param_table_matches_reference_inventory(over):
    from ocrl_amd import _lib
    L = _lib.lib()
    cfg = O.default_cfg(**over)
    c = _lib.SlateConfig(cfg.obs_size, 3, cfg.vocab_size, cfg.d_model, cfg.cnn_hidden, cfg.num_slots, cfg.num_iterations, cfg.slot_size,
                         cfg.mlp_hidden, cfg.num_dec_blocks, cfg.num_dec_heads, cfg.dropout, 2, int(cfg.use_bcdec))
    h = ctypes.c_void_p()
    _lib.check(L.ocrl_slate_create(ctypes.byref(c), ctypes.byref(h)))
    try:
        spec = [(n, s, g) for n, s, g, tr in O.param_shapes(cfg) if tr]
        assert L.ocrl_slate_param_count(h) == len(spec)
        name = ctypes.create_string_buffer(256)
        shape = (ctypes.c_int * 4)()
        nd, off, ne, grp = ctypes.c_int(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_int()
        total, prev_end = 0, 0
        for i, (n, s, g) in enumerate(spec):
            _lib.check(L.ocrl_slate_param_info(h, i, name, 256, ctypes.byref(shape), ctypes.byref(nd), ctypes.byref(off), ctypes.byref(ne), ctypes.byref(grp)))
            assert name.value.decode() == n
            assert tuple(shape[k] for k in range(nd.value)) == tuple(s)
            assert grp.value == g and ne.value == int(np.prod(s))
            assert off.value % 4 == 0 and off.value >= prev_end
            prev_end = off.value + ne.value
            total += ne.value
        assert total == sum(int(np.prod(s)) for _, s, _ in spec)
        assert L.ocrl_slate_flat_size(h) >= total
        gb = [L.ocrl_slate_group_begin(h, g) for g in range(4)]
        assert gb[0] == 0 and gb[0] < gb[1] < gb[2] < gb[3] == L.ocrl_slate_flat_size(h)
        assert L.ocrl_slate_workspace_bytes(h) > 0
    finally:
        L.ocrl_slate_destroy(h)


def test_invalid_config_is_rejected_with_message():
    from ocrl_amd import _lib
    L = _lib.lib()
    c = _lib.SlateConfig(30, 3, 4096, 192, 64, 6, 3, 192, 192, 4, 4, 0.1, 1, 0)      # obs_size not a multiple of 4
    h = ctypes.c_void_p()
    assert L.ocrl_slate_create(ctypes.byref(c), ctypes.byref(h)) != 0
    assert b"invalid" in L.ocrl_last_error()


def test_engine_refuses_cpu_device():
    from ocrl_amd.engine import SlateEngine
    from tests.gpu_util import dims_from_cfg
    with pytest.raises(RuntimeError):
        SlateEngine(dims_from_cfg(O.default_cfg()), 1, device="cpu")


@pytest.mark.parametrize("over", [dict(), dict(obs_size=16, num_slots=3, num_iterations=3, slot_size=32)])
def test_iodine_param_table_matches_reference_inventory(over):
    from ocrl_amd import _lib
    from oracle import iodine_oracle as IO
    L = _lib.lib()
    cfg = IO.default_cfg(**over)
    c = _lib.IodineConfig(cfg.obs_size, 3, cfg.slot_size, cfg.num_iterations, cfg.num_slots, cfg.sigma, cfg.beta, 1, cfg.ref_mlp_hidden, 2)
    h = ctypes.c_void_p()
    _lib.check(L.ocrl_iodine_create(ctypes.byref(c), ctypes.byref(h)))
    try:
        spec = IO.param_shapes(cfg)
        assert L.ocrl_iodine_param_count(h) == len(spec)
        name = ctypes.create_string_buffer(256)
        shape = (ctypes.c_int * 4)()
        nd, off, ne = ctypes.c_int(), ctypes.c_longlong(), ctypes.c_longlong()
        prev_end = 0
        for i, (n, s, _) in enumerate(spec):
            _lib.check(L.ocrl_iodine_param_info(h, i, name, 256, ctypes.byref(shape), ctypes.byref(nd), ctypes.byref(off), ctypes.byref(ne)))
            assert name.value.decode() == n and tuple(shape[k] for k in range(nd.value)) == tuple(s) and ne.value == int(np.prod(s))
            assert off.value % 4 == 0 and off.value >= prev_end
            prev_end = off.value + ne.value
        assert L.ocrl_iodine_flat_size(h) >= prev_end and L.ocrl_iodine_workspace_bytes(h) > 0
    finally:
        L.ocrl_iodine_destroy(h)
    bad = _lib.IodineConfig(40, 3, 64, 5, 7, 0.35, 1.0, 1, 256, 1)           # obs_size not a multiple of 16
    assert L.ocrl_iodine_create(ctypes.byref(bad), ctypes.byref(h)) != 0 and b"invalid" in L.ocrl_last_error()
