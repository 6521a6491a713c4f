"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/orl_engine.h declares.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("orl_build", os.path.join(ROOT, "offlinerl-kit_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(verbose=False)
    from offlinerlkit import _engine
    return _engine.load_library()


def test_header_symbols_exported(lib):
    from offlinerlkit import _engine
    header = open(os.path.join(ROOT, "include", "orl_engine.h")).read()
    declared = set(re.findall(r"\b(orl_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_engine.ABI_SYMBOLS), declared ^ set(_engine.ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), f"missing symbol {sym}"


def test_config_struct_matches_c_layout(lib):
    """orl_config_default fills the struct through the C side: every field lands where ctypes expects it."""
    from offlinerlkit import _engine
    cfg = _engine.default_config("cql")
    assert cfg.algo == 0 and cfg.obs_dim == 17 and cfg.act_dim == 6
    assert cfg.n_hidden == 2 and list(cfg.hidden)[:2] == [256, 256]
    assert cfg.batch_size == 256 and cfg.num_repeat_actions == 10
    assert abs(cfg.cql_weight - 5.0) < 1e-7 and abs(cfg.actor_lr - 1e-4) < 1e-9 and abs(cfg.critic_lr - 3e-4) < 1e-9
    assert abs(cfg.eta - 1.0) < 1e-7 and cfg.num_critics == 10 and cfg.external_arena is None
    # parameter inventory of SURVEY Appendix B: actor 73 484, critic 72 193 (x2 trainable + x2 targets);
    # each net's arena stride is padded to 4 floats (16-B aligned nets -> vector loads)
    assert lib.orl_arena_floats(ctypes.byref(cfg)) == 73484 + 4 * 72196


def test_engine_create_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from offlinerlkit import _engine
    with pytest.raises(RuntimeError, match="HIP device|MI355X"):
        _engine.Engine(_engine.default_config("cql"))
