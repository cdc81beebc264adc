"""The C-ABI library loads on a machine without a GPU and exports every symbol include/*.h declares.
No compute entry point is called here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(inc, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b((?:fs3d|fs2d|fsamd)_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_something():
    names = declared_functions()
    assert "fs3d_advance" in names and "fs3d_create_slab" in names and len(names) > 25


def test_library_exports_every_declared_symbol(fsamd):
    assert os.path.exists(fsamd.LIB_PATH), "build the library first: make -C fluid-simulation_amd/csrc"
    lib = ctypes.CDLL(fsamd.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_binding_covers_every_declared_symbol(fsamd):
    assert sorted(fsamd.EXPORTS) == declared_functions()
    fsamd.load_library()


def test_default_config_is_the_reference_behaviour(fsamd):
    cfg = fsamd.default_config()
    # main.cpp:108: change > 5.0f, iter < 1000; fp32 everywhere (main.cpp:302-314)
    assert (cfg.dtype, cfg.stop_mode, cfg.max_iters, cfg.tol, cfg.omega) == (fsamd.F32, fsamd.STOP_L1_CHANGE, 1000, 5.0, 1.0)
    assert cfg.step_mode == fsamd.STEP_REFERENCE and cfg.local_slabs == 1 and cfg.check_every == 1
    assert ctypes.sizeof(fsamd.Config) == 72 and ctypes.sizeof(fsamd.Stats) == 88 and ctypes.sizeof(fsamd.CommStats) == 72


def test_missing_library_fails_loudly(fsamd, tmp_path):
    import fluid_simulation_amd.binding as b
    saved = b._lib
    b._lib = None
    try:
        with pytest.raises(fsamd.FluidSimError):
            b.load_library(str(tmp_path / "nope.so"))
    finally:
        b._lib = saved


def test_version_string(fsamd):
    assert b"gfx950" in fsamd.load_library().fsamd_version()


def test_rccl_stand_in_exports_what_the_library_binds():
    # the shared-memory transport of tests/fake_rccl must offer every symbol fs_comm.hpp looks up
    import ctypes
    path = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(path):
        pytest.skip("tests/fake_rccl not built")
    lib = ctypes.CDLL(path)
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclGroupStart",
                 "ncclGroupEnd", "ncclAllReduce", "ncclGetErrorString", "ncclCommSplit"):
        assert hasattr(lib, name), name
