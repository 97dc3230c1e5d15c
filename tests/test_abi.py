"""The C-ABI shared library loads and exports every symbol include/relp_engine.h declares
(no compute calls: there is no GPU in the CPU tier)."""
import ctypes
import os
import re

import rust_lp_amd  # noqa: F401
from rust_lp_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "relp_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(relp_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 40
    lib = ctypes.CDLL(engine.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} declared in relp_engine.h but not exported"
        assert name in engine._SIGNATURES, f"{name} has no ctypes signature in rust_lp_amd.engine"
    assert set(engine._SIGNATURES) == set(names)


def test_default_config_matches_reference_defaults():
    cfg = engine.default_config()
    assert cfg.phase_one_rule == engine.FIRST_PROFITABLE_WITH_MEMORY     # phase_one.rs:55,97
    assert cfg.phase_two_rule == engine.STEEPEST_DESCENT                 # two_phase/mod.rs:44
    assert cfg.shard_count == 1 and cfg.device == -1
    assert b"gfx950" in engine.load_library().relp_version()


def test_struct_layouts_match_header():
    # relp_matrix_data_t: 7 int32 (+pad) then pointers; relp_config_t: 3 int32 (+pad), 5 doubles, 10 int32
    assert ctypes.sizeof(engine._MatrixData) == 32 + 8 * 9
    assert ctypes.sizeof(engine.Config) == 16 + 40 + 16 + 8 + 8 + 8


def test_null_handles_are_rejected_not_crashing():
    lib = engine.load_library()
    assert lib.relp_run(None, 1, None, None) == -1
    assert lib.relp_nr_rows(None) == -1
    assert lib.relp_get_b(None, None) == -1
