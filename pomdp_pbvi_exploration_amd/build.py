"""Build the HIP engine in-tree:  python -m pomdp_pbvi_exploration_amd.build

Compiles ``csrc/*.hip`` for gfx950 into ``pomdp_pbvi_exploration_amd/libpbvi_hip.so``
with one explicit ``hipcc`` line (no JIT cache, so the built library travels with
the source tree).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'libpbvi_hip.so')
SOURCES = ['gemm.hip', 'gemm_f64.hip', 'backup_kernels.hip', 'engine.hip']


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(PKG_DIR, '..', 'include', 'pbvi_hip.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', 'hipcc')
    cmd = [hipcc, '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared',
           '-Wall', '-Wno-unused-function', '-pthread',
           *[os.path.join(CSRC, s) for s in SOURCES], '-o', LIB_PATH]
    if verbose:
        print('[build]', ' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB_PATH)
