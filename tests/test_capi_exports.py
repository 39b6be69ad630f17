"""The C-ABI library loads and exports every symbol include/pbvi_hip.h declares (no compute calls:
this runs without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import REPO
from pomdp_pbvi_exploration_amd import engine as eng

HEADER = os.path.join(REPO, 'include', 'pbvi_hip.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pbvi_[a-z0-9_]+)\s*\(', text)))


def test_library_is_built():
    assert os.path.exists(eng.LIB_PATH), 'run `python -m pomdp_pbvi_exploration_amd.build` (or __graft_entry__.build())'


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 15
    lib = ctypes.CDLL(eng.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f'missing exports: {missing}'
    assert sorted(eng.EXPORTS) == names, 'engine.py binds a different set than the header declares'


def test_binding_loads_and_reports_version():
    lib = eng.load_library()
    assert lib.pbvi_version() == 100
    assert lib.pbvi_device_count() >= 0          # 0 here (no GPU), >= 1 on the GPU box


def test_create_fails_loudly_without_gpu():
    lib = eng.load_library()
    if lib.pbvi_device_count() > 0:
        pytest.skip('GPU present')
    import numpy as np
    with pytest.raises(eng.EngineUnavailable):
        eng.Engine(2, 1, 1, 1, np.zeros((2, 1, 1), dtype=np.int64), np.ones((2, 1, 1, 1)), np.zeros((2, 1)))
    h = ctypes.c_void_p()
    rs = (ctypes.c_int32 * 2)(0, 1)
    buf = (ctypes.c_float * 2)(1, 1)
    rc = lib.pbvi_engine_create(ctypes.byref(h), 0, 2, 1, 1, 1, rs, ctypes.cast(buf, ctypes.c_void_p), ctypes.cast(buf, ctypes.c_void_p), 0, 0)
    assert rc == -3 and b'no HIP device' in lib.pbvi_last_error()
    assert lib.pbvi_engine_create(ctypes.byref(h), 0, 0, 1, 1, 1, rs, ctypes.cast(buf, ctypes.c_void_p), ctypes.cast(buf, ctypes.c_void_p), 0, 0) == -1
