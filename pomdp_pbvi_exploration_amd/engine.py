"""ctypes binding of ``include/pbvi_hip.h`` and the ``Engine`` handle wrapper.

This is the only way the package reaches the GPU.  Loading fails loudly
(``EngineUnavailable``) when ``libpbvi_hip.so`` has not been built or cannot be
loaded; there is no CPU stand-in behind it.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, 'libpbvi_hip.so')

PBVI_F32, PBVI_F64 = 0, 1
PBVI_SPARSE, PBVI_DENSE = 0, 1
PBVI_BELIEF_DOMINANCE = 1

EXPORTS = [
    'pbvi_version', 'pbvi_device_count', 'pbvi_last_error', 'pbvi_engine_create', 'pbvi_engine_destroy',
    'pbvi_alpha_set', 'pbvi_alpha_append', 'pbvi_alpha_count', 'pbvi_beliefs_set', 'pbvi_backup_run',
    'pbvi_backup_fetch', 'pbvi_backup_unique_count', 'pbvi_backup_fetch_unique', 'pbvi_backup_device_results',
    'pbvi_backup', 'pbvi_prune_dominated', 'pbvi_value_max',
    'pbvi_set_tie_window', 'pbvi_device_bytes',
    'pbvi_alpha_store_append', 'pbvi_alpha_select', 'pbvi_alpha_store_reset',
    'pbvi_belief_store_append', 'pbvi_beliefs_select', 'pbvi_belief_store_reset', 'pbvi_debug_poison',
    'pbvi_belief_update', 'pbvi_beliefs_advance', 'pbvi_beliefs_fetch', 'pbvi_beliefs_count',
    'pbvi_mdp_value_iteration', 'pbvi_set_formulation', 'pbvi_belief_walk', 'pbvi_engine_set_rto_f64', 'pbvi_backup_fetch_unique_keys', 'pbvi_assemble_rows',
    'pbvi_backup_fetch_exchange', 'pbvi_backup_store_unique',
    'pbvi_value_max_store', 'pbvi_belief_store_count', 'pbvi_alpha_store_count', 'pbvi_set_value_max_exact', 'pbvi_alpha_layout',
    'pbvi_belief_walk_keys', 'pbvi_backup_fetch_value_max',
    'pbvi_backup_fetch_compact', 'pbvi_host_alloc', 'pbvi_host_free', 'pbvi_debug_gemm_dense',
    'pbvi_backup_fetch_exchange_padded', 'pbvi_assemble_rows_store', 'pbvi_exchange_merge', 'pbvi_backup_run_fetch', 'pbvi_debug_alloc_limit', 'pbvi_engine_after_oom', 'pbvi_set_f64_screen', 'pbvi_set_fused_projection', 'pbvi_backup_fetch_row_hashes',
]


class EngineUnavailable(RuntimeError):
    """The HIP library is missing or no GPU is visible."""


class PbviStats(C.Structure):
    _fields_ = [('ms_total', C.c_double), ('ms_project', C.c_double), ('ms_score', C.c_double),
                ('ms_argmax', C.c_double), ('ms_refine', C.c_double), ('ms_action', C.c_double),
                ('ms_assemble', C.c_double), ('ms_dominance', C.c_double), ('n_pairs', C.c_int64),
                ('n_dead', C.c_int64), ('n_refined', C.c_int64), ('n_refined_actions', C.c_int64),
                ('n_unique', C.c_int64),
                ('score_flops', C.c_int64), ('score_flops_executed', C.c_int64), ('score_tiles_dense', C.c_int64),
                ('score_tiles_run', C.c_int64), ('project_flops', C.c_int64), ('project_flops_executed', C.c_int64),
                ('split_k', C.c_int32), ('formulation', C.c_int32), ('n_refine_candidates', C.c_int64),
                ('ms_project_gemm', C.c_double), ('screened', C.c_int32), ('fused_projection', C.c_int32)]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


_lib = None


def load_library(path: str = LIB_PATH):
    """dlopen the engine and declare every prototype of ``include/pbvi_hip.h``."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('PBVI_LIB_PATH', path)         # A/B of experimental builds of the same ABI
    if not os.path.exists(path):
        raise EngineUnavailable(f'{path} not found: build it with `python -m pomdp_pbvi_exploration_amd.build`')
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise EngineUnavailable(f'cannot load {path}: {e}') from e
    vp, i32p, u8p, f64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
    sp = C.POINTER(PbviStats)
    protos = {
        'pbvi_version': (C.c_int, []),
        'pbvi_device_count': (C.c_int, []),
        'pbvi_last_error': (C.c_char_p, []),
        'pbvi_engine_create': (C.c_int, [C.POINTER(vp), C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         i32p, vp, vp, C.c_int, C.c_int]),
        'pbvi_engine_destroy': (None, [vp]),
        'pbvi_alpha_set': (C.c_int, [vp, vp, C.c_int64]),
        'pbvi_alpha_append': (C.c_int, [vp, vp, C.c_int64]),
        'pbvi_alpha_count': (C.c_int64, [vp]),
        'pbvi_beliefs_set': (C.c_int, [vp, vp, C.c_int64]),
        'pbvi_backup_run': (C.c_int, [vp, C.c_double, C.c_int, sp]),
        'pbvi_backup_fetch': (C.c_int, [vp, vp, i32p, i32p, u8p]),
        'pbvi_backup_unique_count': (C.c_int64, [vp]),
        'pbvi_backup_fetch_unique': (C.c_int, [vp, vp, i32p]),
        'pbvi_backup_fetch_compact': (C.c_int, [vp, vp, i32p, i32p, i32p, u8p]),
        'pbvi_host_alloc': (vp, [C.c_size_t]),
        'pbvi_host_free': (None, [vp]),
        'pbvi_debug_gemm_dense': (C.c_int, [C.c_int]),
        'pbvi_backup_store_unique': (C.c_int64, [vp, i32p, C.c_int64]),
        'pbvi_backup_device_results': (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        'pbvi_backup': (C.c_int, [vp, vp, C.c_int64, C.c_double, C.c_int, vp, i32p, i32p, u8p, sp]),
        'pbvi_prune_dominated': (C.c_int, [vp, u8p]),
        'pbvi_value_max': (C.c_int, [vp, f64p, i32p]),
        'pbvi_value_max_store': (C.c_int, [vp, C.c_int64, f64p, i32p]),
        'pbvi_belief_store_count': (C.c_int64, [vp]),
        'pbvi_backup_fetch_value_max': (C.c_int, [vp, f64p]),
        'pbvi_belief_walk_keys': (C.c_int, [vp, C.c_int64, C.POINTER(C.c_uint64)]),
        'pbvi_set_value_max_exact': (C.c_int, [vp, C.c_int]),
        'pbvi_alpha_layout': (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        'pbvi_alpha_store_count': (C.c_int64, [vp]),
        'pbvi_alpha_store_append': (C.c_int64, [vp, vp, C.c_int64]),
        'pbvi_alpha_select': (C.c_int, [vp, i32p, C.c_int64]),
        'pbvi_alpha_store_reset': (C.c_int, [vp]),
        'pbvi_belief_store_append': (C.c_int64, [vp, vp, C.c_int64]),
        'pbvi_beliefs_select': (C.c_int, [vp, i32p, C.c_int64]),
        'pbvi_belief_store_reset': (C.c_int, [vp]),
        'pbvi_debug_poison': (C.c_int, [C.c_int]),
        'pbvi_belief_update': (C.c_int, [vp, i32p, i32p, vp]),
        'pbvi_beliefs_advance': (C.c_int, [vp, i32p, i32p, u8p, C.POINTER(C.c_int64)]),
        'pbvi_beliefs_fetch': (C.c_int, [vp, vp]),
        'pbvi_beliefs_count': (C.c_int64, [vp]),
        'pbvi_mdp_value_iteration': (C.c_int, [C.c_int, C.c_int32, C.c_int32, C.c_int32, i32p, f64p, f64p, f64p,
                                               C.c_double, C.c_double, C.c_int32, f64p, f64p, i32p]),
        'pbvi_set_formulation': (C.c_int, [vp, C.c_int]),
        'pbvi_set_f64_screen': (C.c_int, [vp, C.c_int]),
        'pbvi_set_fused_projection': (C.c_int, [vp, C.c_int]),
        'pbvi_belief_walk': (C.c_int64, [vp, f64p, C.c_int64, i32p, i32p, u8p, f64p]),
        'pbvi_engine_set_rto_f64': (C.c_int, [vp, f64p]),
        'pbvi_backup_fetch_unique_keys': (C.c_int, [vp, vp]),
        'pbvi_backup_fetch_row_hashes': (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        'pbvi_assemble_rows': (C.c_int, [vp, C.c_double, C.c_int64, vp, vp]),
        'pbvi_backup_fetch_exchange': (C.c_int, [vp, vp]),
        'pbvi_backup_fetch_exchange_padded': (C.c_int, [vp, C.c_int64, vp]),
        'pbvi_assemble_rows_store': (C.c_int64, [vp, C.c_double, C.c_int64, vp, vp]),
        'pbvi_backup_run_fetch': (C.c_int, [vp, C.c_double, C.c_int, sp, vp, C.c_int64, i32p, i32p, i32p, i32p, u8p,
                                           C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        'pbvi_debug_alloc_limit': (C.c_int64, [C.c_int64]),
        'pbvi_engine_after_oom': (C.c_int, [vp]),
        'pbvi_exchange_merge': (C.c_int64, [vp, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int64, vp, vp, vp, vp]),
        'pbvi_set_tie_window': (C.c_int, [vp, C.c_double]),
        'pbvi_device_bytes': (C.c_int64, [vp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)          # AttributeError here = the library lost an export
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def debug_poison(enable: bool) -> bool:
    """Fill fresh device allocations with 0xFF (tests); returns the previous setting."""
    return bool(load_library().pbvi_debug_poison(1 if enable else 0))


def debug_alloc_limit(mb: int) -> int:
    """Cap (MiB; < 0: none) on the device bytes one engine may hold (``pbvi_debug_alloc_limit``): a deterministic
    out-of-memory for tests of the ``MemoryError`` contract of ``PBVI_Solver.solve``.  Returns the previous cap."""
    return int(load_library().pbvi_debug_alloc_limit(int(mb)))


def debug_gemm_dense(enable: bool) -> bool:
    """List every GEMM tile, zero or not (``pbvi_debug_gemm_dense``: BASELINE's "dense backup" measurement);
    returns the previous setting."""
    return bool(load_library().pbvi_debug_gemm_dense(1 if enable else 0))


def device_count() -> int:
    return int(load_library().pbvi_device_count())


class _PinnedArray(np.ndarray):
    """An array carved from a ``PinnedBuffer``: it (and every view of it) keeps the buffer alive."""

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, '_owner', None)


class PinnedBuffer:
    """Page-locked host memory from ``pbvi_host_alloc`` viewed as NumPy arrays: results fetched into it are written
    by the GPU's DMA engine directly (no bounce buffer, no CPU copy).  Carved arrays hold a reference to the buffer, so
    it is not freed under them by garbage collection; ``close()`` refuses while any of them is still alive."""

    def __init__(self, nbytes: int):
        self._lib = load_library()
        self.nbytes = int(nbytes)
        self._p = self._lib.pbvi_host_alloc(self.nbytes)
        if not self._p:
            raise MemoryError((self._lib.pbvi_last_error() or b'').decode(errors='replace'))
        self._raw = (C.c_uint8 * self.nbytes).from_address(self._p)
        self._off = 0
        self._views = []

    def carve(self, shape, dtype) -> np.ndarray:
        """Next 256-byte-aligned slice of the buffer as an array of ``shape`` / ``dtype``."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        off = (self._off + 255) // 256 * 256
        if off + n > self.nbytes:
            raise MemoryError('PinnedBuffer exhausted')
        self._off = off + n
        import weakref
        arr = np.frombuffer(self._raw, dtype=dtype, count=int(np.prod(shape)), offset=off).reshape(shape).view(_PinnedArray)
        arr._owner = self
        self._views.append(weakref.ref(arr))
        return arr

    def close(self) -> None:
        if getattr(self, '_p', None):
            if any(r() is not None for r in self._views):
                raise RuntimeError('PinnedBuffer.close(): arrays carved from it are still referenced')
            self._raw = None
            self._lib.pbvi_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mdp_value_iteration(reach_states: np.ndarray, reach_prob: np.ndarray, exp_reward: np.ndarray, v0: np.ndarray,
                        gamma: float, max_change_limit: float, horizon: int, device: int = 0):
    """MDP value-iteration sweeps on the device (``VI_Solver.solve``, ``src/mdp.py:1485-1510``).
    Returns ``(rows [A,S] f64, changes [iterations] f64)``."""
    lib = load_library()
    S, A, R = reach_states.shape
    if int(reach_states.min()) < 0 or int(reach_states.max()) >= S:
        raise ValueError('reachable state out of range')
    rs = np.ascontiguousarray(reach_states, dtype=np.int32)
    p = np.ascontiguousarray(reach_prob, dtype=np.float64)
    er = np.ascontiguousarray(exp_reward, dtype=np.float64)
    v = np.ascontiguousarray(v0, dtype=np.float64)
    if p.shape != (S, A, R) or er.shape != (S, A) or v.shape != (S,):
        raise ValueError('mdp_value_iteration: table shapes do not agree')
    rows = np.empty((A, S), dtype=np.float64)
    changes = np.zeros(max(int(horizon), 1), dtype=np.float64)
    its = C.c_int32(0)
    f64p, i32p = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    _check(lib.pbvi_mdp_value_iteration(int(device), S, A, R, rs.ctypes.data_as(i32p), p.ctypes.data_as(f64p),
                                        er.ctypes.data_as(f64p), v.ctypes.data_as(f64p), float(gamma),
                                        float(max_change_limit), int(horizon), rows.ctypes.data_as(f64p),
                                        changes.ctypes.data_as(f64p), C.byref(its)))
    return rows, changes[:its.value]


def _check(rc: int) -> None:
    if rc == 0:
        return
    msg = (load_library().pbvi_last_error() or b'').decode(errors='replace')
    if rc == -2:
        raise MemoryError(msg)              # PBVI_Solver.solve catches this like the reference (src/pomdp.py:2399)
    if rc == -1:
        raise ValueError(msg)
    if rc == -4:
        raise NotImplementedError(msg)
    raise RuntimeError(f'pbvi engine error {rc}: {msg}')


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class HostArena:
    """Host memory for the arrays the engine hands back and the caller keeps (new alpha rows, walk beliefs), carved
    from a few large blocks instead of one allocation per call.

    Measured on the MI355X box: every time the process maps a fresh region (NumPy does for any array above 128 KiB)
    the next GPU operation can stall 20-40 ms -- the driver revalidates the process's mappings -- even when that
    operation is a copy into pinned memory.  In a solve loop (5 MB of new alpha rows per backup, 24 MB of beliefs per
    expansion) that was 15 stalled backups in 300 and a third of the walks.  With the results living in blocks that
    are mapped once per GiB, the address space does not change between GPU calls."""

    BLOCK_BYTES = 1 << 30               # largest block
    _MIN_BLOCK = 1 << 20
    _POPULATE_MIN = 32 << 20            # smaller blocks are not worth a helper thread
    _POPULATE_CHUNK = 64 << 20
    _MADV_POPULATE_WRITE = 23           # Linux >= 5.14: fault the pages in (writable) without changing their contents

    def __init__(self, first_block_bytes: int = 0):
        """Blocks grow geometrically from ``first_block_bytes`` (a hint: ~100 result rows of the model) up to 1 GiB, so
        a tiger-sized engine holds a MiB, not a GiB, and a row that outlives its engine pins a small block."""
        self._block = None
        self._off = 0
        self._next = max(self._MIN_BLOCK, min(self.BLOCK_BYTES, 1 << max(0, int(first_block_bytes) - 1).bit_length()))

    @classmethod
    def _populate(cls, block: np.ndarray) -> None:
        """Fault a fresh block's pages in from a helper thread.  Untouched, every 4 KiB page of a result costs a
        page fault (and a zeroing) at the moment the engine's copy lands in it -- 0.1 us per KiB measured, 2.4 ms for
        the 24 MB of beliefs one FSVI expansion returns; the helper pays that on another core, ahead of use, and
        ``madvise`` leaves whatever was already written alone."""
        import threading
        try:
            libc = C.CDLL(None, use_errno=True)
            madvise = libc.madvise
        except (OSError, AttributeError):
            return
        madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        madvise.restype = C.c_int
        page = 4096
        lo = (block.ctypes.data + page - 1) // page * page
        hi = (block.ctypes.data + block.shape[0]) // page * page

        def work(keep_alive=block):
            a = lo
            while a < hi:
                n = min(cls._POPULATE_CHUNK, hi - a)
                if madvise(a, n, cls._MADV_POPULATE_WRITE) != 0:
                    return                                   # older kernel / not permitted: pages fault in on first use
                a += n
        threading.Thread(target=work, name='pbvi-arena-populate', daemon=True).start()

    def empty(self, shape, dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        if n == 0 or n > self.BLOCK_BYTES // 4:
            return np.empty(shape, dtype=dtype)              # rare and huge: not worth a block
        off = (self._off + 63) // 64 * 64
        if self._block is None or off + n > self._block.shape[0]:
            size = self._next
            while size < 4 * n:
                size *= 2
            size = min(size, self.BLOCK_BYTES)
            self._next = min(size * 2, self.BLOCK_BYTES)
            self._block = np.empty(size, dtype=np.uint8)
            if size >= self._POPULATE_MIN:
                self._populate(self._block)
            off = 0
        self._off = off + n
        return self._block[off:off + n].view(dtype).reshape(shape)


@dataclass
class BackupResult:
    unique_alpha: np.ndarray    # [U,S] engine dtype: one alpha' row per distinct (a*, v*) key
    index: np.ndarray           # [B] int64: alpha'[b] = unique_alpha[index[b]]
    actions: np.ndarray         # [B] int64
    best_alpha_ind: np.ndarray  # [B,A,O] int64
    keep: np.ndarray            # [B] bool
    stats: dict

    @property
    def alpha(self) -> np.ndarray:
        """Per-belief alpha' matrix [B,S] (what the reference computes before its dedup)."""
        return self.unique_alpha[self.index]

    def value_function_rows(self, use_keep: bool = False, with_index: bool = False):
        """``(rows, actions)`` to hand to ``ValueFunction``: one row per distinct key among the (kept)
        beliefs, in order of first occurrence -- the order the reference's byte-dedup produces.
        ``with_index``: also the positions of those rows in ``unique_alpha`` (for ``Engine.store_unique``)."""
        idx = self.index[self.keep] if use_keep else self.index
        act = self.actions[self.keep] if use_keep else self.actions
        if idx.size == 0:
            return (self.unique_alpha[:0], act[:0], idx[:0]) if with_index else (self.unique_alpha[:0], act[:0])
        first = np.unique(idx, return_index=True)[1]
        first.sort()
        sel = idx[first]
        # the engine already lists the distinct rows in order of first occurrence: unless the belief-dominance mask dropped
        # some, `sel` is 0 .. U-1 and the rows are handed on as they lie (a fancy-index copy of U rows of 120-240 KB was
        # 0.7 ms of every backup of a solve loop)
        whole = len(sel) == len(self.unique_alpha) and bool((sel == np.arange(len(sel))).all())
        rows = self.unique_alpha if whole else self.unique_alpha[sel]
        if with_index:
            return rows, act[first], sel
        return rows, act[first]


class Engine:
    """One model on one GPU: resident tables, alpha set and belief block."""

    _serials = iter(range(1, 1 << 62))      # process-wide: a residency tag must never match a dead engine's (ids are reused)
    _nonce = None                           # ... nor an engine's of another process or run (pickled / copied objects keep
                                            # their tags): the serial carries (pid, 48 random bits drawn once per process)

    def __init__(self, S: int, A: int, O: int, R: int, reach_states: np.ndarray, rto: np.ndarray,
                 exp_rewards: np.ndarray, dtype: str = 'f32', mode: str = 'sparse', device: int = 0):
        lib = load_library()
        if lib.pbvi_device_count() <= 0:
            raise EngineUnavailable('no HIP device visible to libpbvi_hip.so')
        assert dtype in ('f32', 'f64')
        self.dtype = dtype
        self.np_dtype = np.float32 if dtype == 'f32' else np.float64
        self.S, self.A, self.O, self.R = int(S), int(A), int(O), int(R)
        self.device = device
        rs = np.asarray(reach_states)
        assert rs.shape == (S, A, R), f'reachable_states must be [S,A,R]={S, A, R}, got {rs.shape}'
        if rs.size and (rs.min() < 0 or rs.max() >= S):
            raise ValueError('reachable_states entry out of range [0,S)')
        rs32 = np.ascontiguousarray(rs, dtype=np.int32)
        rto_c = np.ascontiguousarray(rto, dtype=self.np_dtype)
        er_c = np.ascontiguousarray(exp_rewards, dtype=self.np_dtype)
        assert rto_c.shape == (S, A, O, R) and er_c.shape == (S, A)
        self._h = C.c_void_p()
        _check(lib.pbvi_engine_create(C.byref(self._h), device, S, A, O, R,
                                      rs32.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(rto_c), _ptr(er_c),
                                      PBVI_F32 if dtype == 'f32' else PBVI_F64,
                                      PBVI_SPARSE if mode == 'sparse' else PBVI_DENSE))
        self._lib = lib
        if Engine._nonce is None or Engine._nonce[0] != os.getpid():
            import secrets
            Engine._nonce = (os.getpid(), secrets.randbits(48))
        self.serial = Engine._nonce + (next(Engine._serials),)
        self._alpha_token = None
        self._formulation = None                # see the property
        self.last_stats = {}                    # pbvi_stats_t of the last run()
        self._store_epoch = {'alpha': 0, 'belief': 0}
        self._resident = {'alpha': None, 'belief': None}     # store ids of the working alpha set / belief block
        self.B = 0
        self._vmax_cache, self._vmax_epochs = [], None
        self._arena = HostArena(first_block_bytes=128 * self.S * 8)
        if dtype == 'f32' and np.asarray(rto).dtype == np.float64:
            # the belief walk returns fp64 belief values to the host containers: keep them independent of the
            # engine's arithmetic type by giving it the fp64 table too (a few MB)
            r64 = np.ascontiguousarray(rto, dtype=np.float64)
            self._ck(lib.pbvi_engine_set_rto_f64(self._h, r64.ctypes.data_as(C.POINTER(C.c_double))))

    @classmethod
    def for_model(cls, model, dtype: str = 'f64', device: int = 0, mode: str = 'sparse') -> 'Engine':
        """Engine over a host ``pomdp.Model`` (what ``Model.gpu_model`` does in the reference)."""
        return cls(model.state_count, model.action_count, model.observation_count, model.reachable_state_count,
                   model.reachable_states, model.reachable_transitional_observation_table,
                   model.expected_rewards_table, dtype=dtype, mode=mode, device=device)

    def close(self) -> None:
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.pbvi_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc: int) -> None:
        """``_check`` for calls on this engine: a device allocation failure (-2) first returns the engine to its freshly
        created state (``pbvi_engine_after_oom``: working sets, row stores and scratch released) and forgets everything
        this wrapper cached about device residency, then raises ``MemoryError`` -- which ``PBVI_Solver.solve`` turns into
        "return the partial result" like the reference (``src/pomdp.py:2399-2401``)."""
        if rc == -2:
            msg = (self._lib.pbvi_last_error() or b'').decode(errors='replace')
            self._lib.pbvi_engine_after_oom(self._h)
            self._resident = {'alpha': None, 'belief': None}
            for k in self._store_epoch:                  # residency tags of AlphaVector / Belief objects no longer match
                self._store_epoch[k] += 1
            self._vmax_cache, self._vmax_epochs = [], None
            self._alpha_token = None
            self.B = 0
            raise MemoryError(msg)
        _check(rc)

    # -- residency ------------------------------------------------------- #
    def _as_rows(self, arr: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(arr, dtype=self.np_dtype)
        if a.ndim != 2 or a.shape[1] != self.S:
            raise ValueError(f'expected a [*, {self.S}] array, got {a.shape}')
        return a

    def set_alpha(self, alpha: np.ndarray) -> None:
        a = self._as_rows(alpha)
        self._resident['alpha'] = None
        self._ck(self._lib.pbvi_alpha_set(self._h, _ptr(a), a.shape[0]))
        self._alpha_token = None

    def append_alpha(self, alpha: np.ndarray) -> None:
        a = self._as_rows(alpha)
        self._resident['alpha'] = None
        self._ck(self._lib.pbvi_alpha_append(self._h, _ptr(a), a.shape[0]))
        self._alpha_token = None

    @property
    def alpha_count(self) -> int:
        return int(self._lib.pbvi_alpha_count(self._h))

    def set_beliefs(self, beliefs: np.ndarray) -> None:
        b = self._as_rows(beliefs)
        self._resident['belief'] = None
        self._ck(self._lib.pbvi_beliefs_set(self._h, _ptr(b), b.shape[0]))
        self.B = b.shape[0]

    # -- device row stores: upload once, select by id in host order ------- #
    def store_rows(self, which: str, rows: np.ndarray) -> int:
        """Append ``rows`` [n,S] to the alpha ('alpha') or belief ('belief') store; returns the first id."""
        a = self._as_rows(rows)
        fn = self._lib.pbvi_alpha_store_append if which == 'alpha' else self._lib.pbvi_belief_store_append
        first = int(fn(self._h, _ptr(a), a.shape[0]))
        if first < 0:
            self._ck(first)
        return first

    def store_unique(self, unique_idx) -> int:
        """Append rows ``unique_idx`` of the last backup's distinct alpha' rows to the alpha store, device to device
        (``pbvi_backup_store_unique``); returns the store id of the first one."""
        i = np.ascontiguousarray(unique_idx, dtype=np.int32)
        first = int(self._lib.pbvi_backup_store_unique(self._h, i.ctypes.data_as(C.POINTER(C.c_int32)), i.shape[0]))
        if first < 0:
            self._ck(first)
        return first

    def store_tag(self, which: str):
        """What ``row_ids`` expects in ``obj._dev[0]`` for a row of this engine's store."""
        return (self.serial, which, self._store_epoch[which])

    # The working alpha set / belief block is a gathered copy of store rows (1.2 GB for 10^4 alpha rows): selecting
    # the ids that are already resident is skipped.  `_resident[which]` is dropped by everything else that rewrites
    # the working set (set_*, append_alpha, advance_beliefs, belief updates, store resets).
    def select_alpha(self, ids) -> None:
        i = np.ascontiguousarray(ids, dtype=np.int32)
        have = self._resident['alpha']
        if have is not None and have.shape == i.shape and np.array_equal(have, i):
            return
        self._resident['alpha'] = None
        self._ck(self._lib.pbvi_alpha_select(self._h, i.ctypes.data_as(C.POINTER(C.c_int32)), i.shape[0]))
        self._resident['alpha'] = i.copy()

    def select_beliefs(self, ids) -> None:
        i = np.ascontiguousarray(ids, dtype=np.int32)
        have = self._resident['belief']
        if have is not None and have.shape == i.shape and np.array_equal(have, i):
            return
        self._resident['belief'] = None
        self._ck(self._lib.pbvi_beliefs_select(self._h, i.ctypes.data_as(C.POINTER(C.c_int32)), i.shape[0]))
        self.B = i.shape[0]
        self._resident['belief'] = i.copy()

    def alpha_layout(self):
        """``(extendable, free_rows, layouts)`` of the working alpha set (``pbvi_alpha_layout``): whether it is a store
        selection that a new-then-old extension can grow at the front, the rows still free there, and how many times a
        selection was gathered afresh so far."""
        free, lay = C.c_int64(0), C.c_int64(0)
        rc = self._lib.pbvi_alpha_layout(self._h, C.byref(free), C.byref(lay))
        if rc < 0:
            self._ck(rc)
        return bool(rc), int(free.value), int(lay.value)

    def reset_store(self, which: str) -> None:
        self._resident[which] = None
        self._ck((self._lib.pbvi_alpha_store_reset if which == 'alpha' else self._lib.pbvi_belief_store_reset)(self._h))
        self._store_epoch[which] += 1

    def row_ids(self, which: str, objects, values_of, owner=None) -> np.ndarray:
        """Store ids of ``objects`` (AlphaVector / Belief instances, list order); rows not yet in this engine's
        store are uploaded in one batch.  The id lives on the object (``_dev``), tagged with this engine and the
        store epoch.  ``owner`` (the ValueFunction / BeliefSet holding the list) caches the id array so the
        per-object walk happens once per container, not once per call; containers drop ``_dev_ids`` when they
        change."""
        tag = (self.serial, which, self._store_epoch[which])
        if owner is not None:
            c = getattr(owner, '_dev_ids', None)
            if c is not None and c[0] == tag and len(c[1]) == len(objects):
                return c[1]
        missing = [o for o in objects if getattr(o, '_dev', (None, -1))[0] != tag]
        if missing:
            # the same object may appear twice in a list: upload it once
            seen, todo = set(), []
            for o in missing:
                if id(o) not in seen:
                    seen.add(id(o))
                    todo.append(o)
            first = self.store_rows(which, np.stack([np.asarray(values_of(o)) for o in todo]))
            for k, o in enumerate(todo):
                o._dev = (tag, first + k)
        ids = np.fromiter((o._dev[1] for o in objects), dtype=np.int32, count=len(objects))
        if owner is not None:
            owner._dev_ids = (tag, ids)
        return ids

    def sync_rows(self, which: str, objects, values_of, owner=None) -> None:
        """Make ``objects`` the working set: upload what is missing (``row_ids``), then select by id."""
        ids = self.row_ids(which, objects, values_of, owner)
        if which == 'alpha':
            self.select_alpha(ids)
        else:
            self.select_beliefs(ids)

    # -- max_v b.alpha_v with memory (compute_change on a growing belief set / alpha set) -------------- #
    _VMAX_ENTRIES = 3
    _BLOCK = 32768          # beliefs per resident block (the engine takes at most 65535)

    def max_value_store(self, n: int = -1):
        """``(max_v b.alpha_v, argmax)`` for belief-store rows ``[0, n)`` (all rows by default) against the working
        alpha set, computed on the store in place (``pbvi_value_max_store``)."""
        if n < 0:
            n = int(self._lib.pbvi_belief_store_count(self._h))
        val = np.empty(n, dtype=np.float64)
        idx = np.empty(n, dtype=np.int32)
        self._ck(self._lib.pbvi_value_max_store(self._h, n, val.ctypes.data_as(C.POINTER(C.c_double)),
                                              idx.ctypes.data_as(C.POINTER(C.c_int32))))
        return val, idx

    _STORE_SCAN_MIN = 2048      # beliefs from which scoring the store in place beats gathering a block

    def _vmax_block(self, a_ids: np.ndarray, b_ids: np.ndarray) -> np.ndarray:
        out = np.empty(len(b_ids), dtype=np.float64)
        self.select_alpha(a_ids)
        if len(b_ids) >= self._STORE_SCAN_MIN:
            top = int(b_ids.max()) + 1
            if 2 * len(b_ids) >= top:          # most of the store's prefix is wanted: score it whole, pick by id
                return self.max_value_store(top)[0][b_ids]
        for i0 in range(0, len(b_ids), self._BLOCK):
            self.select_beliefs(b_ids[i0:i0 + self._BLOCK])
            out[i0:i0 + self._BLOCK] = self.max_value_resident()[0]
        return out

    def best_alpha_of_store_rows(self, b_ids: np.ndarray) -> np.ndarray:
        """``argmax_v b.alpha_v`` (first maximum, exact) of belief-store rows ``b_ids`` against the working alpha set: the
        usefulness scan of the solve loop's |V| limiter (``src/pomdp.py:2350-2353``).  Most of the store's prefix wanted
        (the solve loop: all of it): scored in place; else in gathered blocks."""
        b_ids = np.asarray(b_ids, dtype=np.int32)
        if len(b_ids) == 0:
            return np.zeros(0, dtype=np.int64)
        top = int(b_ids.max()) + 1
        if 2 * len(b_ids) >= top:
            return self.max_value_store(top)[1][b_ids].astype(np.int64)
        out = np.empty(len(b_ids), dtype=np.int64)
        for i0 in range(0, len(b_ids), self._BLOCK):
            self.select_beliefs(b_ids[i0:i0 + self._BLOCK])
            out[i0:i0 + self._BLOCK] = self.max_value_resident()[1]
        return out

    def seed_max_values(self, alpha_objects, belief_objects, alpha_values, belief_values, alpha_owner=None,
                        belief_owner=None) -> bool:
        """After a backup in the belief-side formulation: take ``max_v b.alpha_v`` of its beliefs against its alpha set
        from the engine (``pbvi_backup_fetch_value_max`` -- the beliefs rode along in the score GEMM) and put it where
        ``max_value_objects`` will look, so ``compute_change`` does not run that GEMM again.  Returns False when the
        last backup has no such values (alpha-side formulation)."""
        vals = np.empty(self.B, dtype=np.float64)
        rc = self._lib.pbvi_backup_fetch_value_max(self._h, vals.ctypes.data_as(C.POINTER(C.c_double)))
        if rc == -4:
            return False
        self._ck(rc)
        a_ids = self.row_ids('alpha', alpha_objects, alpha_values, alpha_owner)
        b_ids = self.row_ids('belief', belief_objects, belief_values, belief_owner)
        if len(b_ids) != len(vals):
            return False
        epochs = (self._store_epoch['alpha'], self._store_epoch['belief'])
        if self._vmax_epochs != epochs:
            self._vmax_cache, self._vmax_epochs = [], epochs
        exact = self.dtype != 'f32'             # fp32 engines: the GEMM's maxima, the exact=False pool of max_value_objects
        aset = np.unique(a_ids)
        hit = next((e for e in self._vmax_cache if e['exact'] == exact and np.array_equal(e['aset'], aset)), None)
        n_ids = int(b_ids.max()) + 1
        if hit is None:
            hit = {'aset': aset, 'vals': np.full(n_ids, np.nan), 'exact': exact}
            self._vmax_cache.append(hit)
            del self._vmax_cache[:-self._VMAX_ENTRIES]
        elif len(hit['vals']) < n_ids:
            hit['vals'] = np.concatenate([hit['vals'], np.full(n_ids - len(hit['vals']), np.nan)])
        unknown = np.isnan(hit['vals'][b_ids])
        hit['vals'][b_ids[unknown]] = vals[unknown]        # a value scored before stays (it is the same sum)
        return True

    def set_value_max_exact(self, exact: bool) -> None:
        """f32 engines: fp64 re-scoring of ``max_value_*`` results on (default) or off (``pbvi_set_value_max_exact``)."""
        self._ck(self._lib.pbvi_set_value_max_exact(self._h, 1 if exact else 0))

    def max_value_objects(self, alpha_objects, belief_objects, alpha_values, belief_values, alpha_owner=None,
                          belief_owner=None, exact: bool = True) -> np.ndarray:
        if not exact and self.dtype == 'f32':
            self.set_value_max_exact(False)
            try:
                return self._max_value_objects(alpha_objects, belief_objects, alpha_values, belief_values, alpha_owner,
                                               belief_owner, False)
            finally:
                self.set_value_max_exact(True)
        return self._max_value_objects(alpha_objects, belief_objects, alpha_values, belief_values, alpha_owner,
                                       belief_owner, True)

    def _max_value_objects(self, alpha_objects, belief_objects, alpha_values, belief_values, alpha_owner, belief_owner,
                           exact: bool) -> np.ndarray:
        """``max_v b.alpha_v`` for every belief object against the set of alpha objects, reusing earlier results.

        ``compute_change`` (``src/pomdp.py:2141-2169``) asks for this twice per backup on the whole accumulated
        belief set.  The maximum over a set that grew is the maximum of the old value and the maximum over the new
        rows, and a belief scored before keeps its value, so only (known beliefs x new alpha rows) and
        (new beliefs x all alpha rows) go through the GEMM.  Values are the engine's exact re-scored ones, so the
        result equals the from-scratch one.  Entries are keyed by the set of alpha store ids (and by whether the values
        are exact: ``exact=False`` -- f32 engines, ``pbvi_set_value_max_exact`` -- keeps the fp32 GEMM's maxima)."""
        a_ids = self.row_ids('alpha', alpha_objects, alpha_values, alpha_owner)
        b_ids = self.row_ids('belief', belief_objects, belief_values, belief_owner)
        epochs = (self._store_epoch['alpha'], self._store_epoch['belief'])
        if self._vmax_epochs != epochs:
            self._vmax_cache, self._vmax_epochs = [], epochs
        aset = np.unique(a_ids)                             # sorted ids: set algebra stays in NumPy
        pool = [e for e in self._vmax_cache if e['exact'] == exact]
        hit = next((e for e in pool if np.array_equal(e['aset'], aset)), None)
        base = hit
        if base is None:     # largest cached subset of this alpha set
            subs = [e for e in pool
                    if len(e['aset']) <= len(aset) and np.isin(e['aset'], aset, assume_unique=True).all()]
            base = max(subs, key=lambda e: len(e['aset'])) if subs else None
        n_ids = int(b_ids.max()) + 1 if len(b_ids) else 0
        vals = np.full(len(b_ids), np.nan)
        if base is not None:
            have = base['vals']
            inside = b_ids < len(have)
            vals[inside] = have[b_ids[inside]]
        known = ~np.isnan(vals)
        if base is not None and base is not hit and known.any():
            fresh = np.setdiff1d(aset, base['aset'], assume_unique=True).astype(np.int32)
            if len(fresh):
                vals[known] = np.maximum(vals[known], self._vmax_block(fresh, b_ids[known]))
        if (~known).any():
            vals[~known] = self._vmax_block(a_ids, b_ids[~known])
        if hit is None:
            hit = {'aset': aset, 'vals': np.full(n_ids, np.nan), 'exact': exact}
            self._vmax_cache.append(hit)
        elif len(hit['vals']) < n_ids:
            hit['vals'] = np.concatenate([hit['vals'], np.full(n_ids - len(hit['vals']), np.nan)])
        hit['vals'][b_ids] = vals
        self._vmax_cache = [c for c in self._vmax_cache if c is not hit] + [hit]    # most recent last
        del self._vmax_cache[:-self._VMAX_ENTRIES]
        return vals

    def _ensure_alpha(self, alpha: np.ndarray) -> None:
        """Make ``alpha`` the resident set.  Always uploads: array identity is not a safe
        cache key (in-place edits, reused addresses); callers that want residency across
        calls use ``set_alpha`` / ``append_alpha`` + ``run`` directly."""
        self.set_alpha(alpha)

    # -- the backup ------------------------------------------------------ #
    def run(self, gamma: float, belief_dominance_prune: bool = False) -> dict:
        """Backup of the resident belief block against the resident alpha set; results stay on the device."""
        st = PbviStats()
        self._ck(self._lib.pbvi_backup_run(self._h, float(gamma), PBVI_BELIEF_DOMINANCE if belief_dominance_prune else 0,
                                         C.byref(st)))
        self.last_stats = st.as_dict()
        return self.last_stats

    def fetch(self) -> BackupResult:
        """Results of the last run: unique alpha' rows + per-belief index (the D2H copy moves U rows, not B)."""
        B = self.B
        U = int(self._lib.pbvi_backup_unique_count(self._h))
        if U < 0:
            raise ValueError('no backup result resident')
        rows = self._arena.empty((U, self.S), self.np_dtype)      # kept by the caller as AlphaVector values
        index = np.empty(B, dtype=np.int32)
        act = np.empty(B, dtype=np.int32)
        best = np.empty((B, self.A, self.O), dtype=np.int32)
        keep = np.empty(B, dtype=np.uint8)
        i32p = C.POINTER(C.c_int32)
        self._ck(self._lib.pbvi_backup_fetch_compact(self._h, _ptr(rows), index.ctypes.data_as(i32p), act.ctypes.data_as(i32p),
                                                   best.ctypes.data_as(i32p), keep.ctypes.data_as(C.POINTER(C.c_uint8))))
        return BackupResult(rows, index.astype(np.int64), act.astype(np.int64), best.astype(np.int64),
                            keep.astype(bool), {})

    def fetch_compact_into(self, rows: np.ndarray, index: np.ndarray, actions: np.ndarray, best=None, keep=None) -> int:
        """Results of the last run into caller-owned arrays (``pbvi_backup_fetch_compact``; pinned arrays -- see
        ``PinnedBuffer`` -- are written by DMA directly): the U distinct rows into ``rows[:U]``, ``index`` [B] int32,
        ``actions`` [B] int32, optionally ``best`` [B,A,O] int32 and ``keep`` [B] uint8.  One synchronisation; returns U."""
        U = int(self._lib.pbvi_backup_unique_count(self._h))
        if U < 0:
            raise ValueError('no backup result resident')
        if rows.shape[0] < U or rows.shape[1] != self.S or rows.dtype != self.np_dtype or not rows.flags.c_contiguous:
            raise ValueError(f'rows must be a C-contiguous [>= {U}, {self.S}] {self.dtype} array')
        for name, a, shape, dt in (('index', index, (self.B,), np.int32), ('actions', actions, (self.B,), np.int32),
                                   ('best', best, (self.B, self.A, self.O), np.int32), ('keep', keep, (self.B,), np.uint8)):
            if a is not None and (a.shape != shape or a.dtype != dt or not a.flags.c_contiguous):
                raise ValueError(f'{name} must be a C-contiguous {shape} {np.dtype(dt).name} array')
        i32p = C.POINTER(C.c_int32)
        self._ck(self._lib.pbvi_backup_fetch_compact(
            self._h, _ptr(rows), index.ctypes.data_as(i32p), actions.ctypes.data_as(i32p),
            best.ctypes.data_as(i32p) if best is not None else None,
            keep.ctypes.data_as(C.POINTER(C.c_uint8)) if keep is not None else None))
        return U

    def run_fetch_into(self, gamma: float, rows: np.ndarray, slot: np.ndarray, index: np.ndarray, actions: np.ndarray,
                       best=None, keep=None, belief_dominance_prune: bool = False):
        """``run`` + ``fetch_compact_into`` in one call with the rows leaving early (``pbvi_backup_run_fetch``): ``rows`` must be
        page-locked (``PinnedBuffer``) with room for B rows; the row of distinct key ``u`` is ``rows[slot[u]]``, and
        ``alpha'[b] == rows[slot[index[b]]]``.  Returns ``(stats, U, slots_used)``."""
        if rows.shape[0] < self.B or rows.shape[1] != self.S or rows.dtype != self.np_dtype or not rows.flags.c_contiguous:
            raise ValueError(f'rows must be a C-contiguous [>= {self.B}, {self.S}] {self.dtype} array')
        for name, a, shape, dt in (('slot', slot, (self.B,), np.int32), ('index', index, (self.B,), np.int32),
                                   ('actions', actions, (self.B,), np.int32), ('best', best, (self.B, self.A, self.O), np.int32),
                                   ('keep', keep, (self.B,), np.uint8)):
            if a is not None and (a.shape != shape or a.dtype != dt or not a.flags.c_contiguous):
                raise ValueError(f'{name} must be a C-contiguous {shape} {np.dtype(dt).name} array')
        i32p = C.POINTER(C.c_int32)
        st = PbviStats()
        nu, ns = C.c_int64(0), C.c_int64(0)
        self._ck(self._lib.pbvi_backup_run_fetch(
            self._h, float(gamma), PBVI_BELIEF_DOMINANCE if belief_dominance_prune else 0, C.byref(st), _ptr(rows), rows.shape[0],
            slot.ctypes.data_as(i32p), index.ctypes.data_as(i32p), actions.ctypes.data_as(i32p),
            best.ctypes.data_as(i32p) if best is not None else None,
            keep.ctypes.data_as(C.POINTER(C.c_uint8)) if keep is not None else None, C.byref(nu), C.byref(ns)))
        return st.as_dict(), int(nu.value), int(ns.value)

    def fetch_full(self, out=None) -> np.ndarray:
        """Per-belief alpha' matrix [B,S] expanded on the device (``pbvi_backup_fetch``'s out_alpha); ``out``: a
        C-contiguous [B,S] array of the engine's dtype to fill instead of a new one."""
        if out is None:
            out = np.empty((self.B, self.S), dtype=self.np_dtype)
        elif out.shape != (self.B, self.S) or out.dtype != self.np_dtype or not out.flags.c_contiguous:
            raise ValueError(f'out must be a C-contiguous [{self.B}, {self.S}] {self.dtype} array')
        self._ck(self._lib.pbvi_backup_fetch(self._h, _ptr(out), None, None, None))
        return out

    @property
    def unique_count(self) -> int:
        return int(self._lib.pbvi_backup_unique_count(self._h))

    def fetch_unique_into(self, rows_ptr: int, index_ptr: int) -> None:
        """Copy unique rows [U,S] / index [B] to raw (host or device) addresses."""
        self._ck(self._lib.pbvi_backup_fetch_unique(self._h, C.c_void_p(rows_ptr) if rows_ptr else None,
                                                  C.cast(index_ptr, C.POINTER(C.c_int32)) if index_ptr else None))

    def fetch_row_hashes(self) -> np.ndarray:
        """``[U]`` uint64: the position-weighted bit-pattern hashes of the last backup's distinct rows, computed on the
        device (``pbvi_backup_fetch_row_hashes``); ``mdp._AlphaKey`` is the host's function of the same name."""
        U = self.unique_count
        out = np.empty(max(U, 0), dtype=np.uint64)
        if U > 0:
            self._ck(self._lib.pbvi_backup_fetch_row_hashes(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def fetch_unique_keys(self) -> np.ndarray:
        """``[U, 1+O]`` int32: (a*, v*[a*, :]) of each distinct alpha' row of the last backup."""
        U = self.unique_count
        keys = np.empty((U, 1 + self.O), dtype=np.int32)
        if U:
            self._ck(self._lib.pbvi_backup_fetch_unique_keys(self._h, _ptr(keys)))
        return keys

    def fetch_unique_keys_into(self, keys_ptr: int) -> None:
        self._ck(self._lib.pbvi_backup_fetch_unique_keys(self._h, C.c_void_p(keys_ptr)))

    def fetch_exchange_into(self, ptr: int, per: int = None) -> None:
        """``[U | index[per] | actions[per] | keep[per] | keys[per][1+O]]`` int32 at a raw (host or device) address;
        ``per`` (default B) >= B is the common block size of a sharded run."""
        if per is None:
            self._ck(self._lib.pbvi_backup_fetch_exchange(self._h, C.c_void_p(ptr)))
        else:
            self._ck(self._lib.pbvi_backup_fetch_exchange_padded(self._h, int(per), C.c_void_p(ptr)))

    def exchange_size(self, per: int) -> int:
        """int32 entries of one rank's exchange message for block size ``per``."""
        return 1 + 3 * per + per * (1 + self.O)

    def assemble_rows_store(self, keys: np.ndarray, gamma: float, want_rows: bool = True):
        """alpha' rows for ``keys [n, 1+O]`` against the resident alpha set, appended to the alpha store device to
        device (``pbvi_assemble_rows_store``).  Returns ``(rows [n,S] or None, first store id)``."""
        k = np.ascontiguousarray(keys, dtype=np.int32)
        if k.ndim != 2 or k.shape[1] != 1 + self.O or k.shape[0] == 0:
            raise ValueError(f'keys must be [n >= 1, {1 + self.O}]')
        out = self._arena.empty((k.shape[0], self.S), self.np_dtype) if want_rows else None
        first = int(self._lib.pbvi_assemble_rows_store(self._h, float(gamma), k.shape[0], _ptr(k),
                                                       _ptr(out) if out is not None else None))
        if first < 0:
            self._ck(first)
        return out, first

    def assemble_rows_store_from(self, keys_ptr: int, n: int, gamma: float) -> int:
        """Same for keys at a raw (host or device) address; rows stay on the device.  Returns the first store id."""
        first = int(self._lib.pbvi_assemble_rows_store(self._h, float(gamma), int(n), C.c_void_p(keys_ptr), None))
        if first < 0:
            self._ck(first)
        return first

    def assemble_rows(self, keys: np.ndarray, gamma: float) -> np.ndarray:
        """alpha' rows ``[n, S]`` for ``keys [n, 1+O]`` against the resident alpha set (``pbvi_assemble_rows``)."""
        k = np.ascontiguousarray(keys, dtype=np.int32)
        if k.ndim != 2 or k.shape[1] != 1 + self.O:
            raise ValueError(f'keys must be [n, {1 + self.O}]')
        out = np.empty((k.shape[0], self.S), dtype=self.np_dtype)
        if k.shape[0]:
            self._ck(self._lib.pbvi_assemble_rows(self._h, float(gamma), k.shape[0], _ptr(k), _ptr(out)))
        return out

    def assemble_rows_into(self, keys_ptr: int, n: int, gamma: float, out_ptr: int) -> None:
        self._ck(self._lib.pbvi_assemble_rows(self._h, float(gamma), int(n), C.c_void_p(keys_ptr), C.c_void_p(out_ptr)))

    def fetch_into(self, alpha_ptr: int, action_ptr: int, keep_ptr: int) -> None:
        """Copy the last run's alpha rows / actions / keep mask to raw addresses (host or
        device memory of this GPU), e.g. the ``data_ptr()`` of the RCCL send buffers."""
        self._ck(self._lib.pbvi_backup_fetch(self._h, C.c_void_p(alpha_ptr) if alpha_ptr else None,
                                           C.cast(action_ptr, C.POINTER(C.c_int32)) if action_ptr else None, None,
                                           C.cast(keep_ptr, C.POINTER(C.c_uint8)) if keep_ptr else None))

    def backup_full(self, alpha: np.ndarray, beliefs: np.ndarray, gamma: float,
                    belief_dominance_prune: bool = False) -> BackupResult:
        self._ensure_alpha(alpha)
        self.set_beliefs(beliefs)
        stats = self.run(gamma, belief_dominance_prune)
        res = self.fetch()
        res.stats = stats
        return res

    def backup(self, alpha: np.ndarray, beliefs: np.ndarray, gamma: float, belief_dominance_prune: bool = False):
        """``(rows, actions)`` ready for ``ValueFunction(model, rows, actions)``: the new alpha-vectors of the
        (dominating) beliefs, already reduced to one row per distinct key in first-occurrence order."""
        res = self.backup_full(alpha, beliefs, gamma, belief_dominance_prune)
        return res.value_function_rows(use_keep=belief_dominance_prune)

    def device_results(self):
        """Raw device addresses ``(alpha_ptr, action_ptr, keep_ptr)`` of the last run (for RCCL)."""
        a, c, k = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._ck(self._lib.pbvi_backup_device_results(self._h, C.byref(a), C.byref(c), C.byref(k)))
        return a.value, c.value, k.value

    # -- companions of the backup --------------------------------------- #
    def prune_dominated(self, alpha: np.ndarray) -> np.ndarray:
        self._ensure_alpha(alpha)
        keep = np.empty(alpha.shape[0], dtype=np.uint8)
        self._ck(self._lib.pbvi_prune_dominated(self._h, keep.ctypes.data_as(C.POINTER(C.c_uint8))))
        return keep.astype(bool)

    def prune_dominated_objects(self, objects, values_of, owner=None) -> np.ndarray:
        """Same for a list of AlphaVector objects: rows already in the device store are not uploaded again."""
        self.sync_rows('alpha', objects, values_of, owner)
        keep = np.empty(len(objects), dtype=np.uint8)
        self._ck(self._lib.pbvi_prune_dominated(self._h, keep.ctypes.data_as(C.POINTER(C.c_uint8))))
        return keep.astype(bool)

    def max_value(self, alpha: np.ndarray, beliefs: np.ndarray):
        """``(max_v b.alpha_v [B] f64, argmax [B])`` (compute_change, ``src/pomdp.py:2165``)."""
        self._ensure_alpha(alpha)
        self.set_beliefs(beliefs)
        return self.max_value_resident()

    def max_value_resident(self):
        """Same, for the working alpha set / belief block already selected on the device."""
        val = np.empty(self.B, dtype=np.float64)
        idx = np.empty(self.B, dtype=np.int32)
        self._ck(self._lib.pbvi_value_max(self._h, val.ctypes.data_as(C.POINTER(C.c_double)),
                                        idx.ctypes.data_as(C.POINTER(C.c_int32))))
        return val, idx.astype(np.int64)

    def belief_update(self, beliefs: np.ndarray, actions, observations) -> np.ndarray:
        """Batched Bayes step: row b of the result is ``Belief(beliefs[b]).update(actions[b], observations[b])``
        (``src/pomdp.py:382-421``) computed on the device."""
        self.set_beliefs(beliefs)
        a = np.ascontiguousarray(actions, dtype=np.int32)
        o = np.ascontiguousarray(observations, dtype=np.int32)
        if a.shape != (self.B,) or o.shape != (self.B,):
            raise ValueError('actions / observations must be [B]')
        out = np.empty((self.B, self.S), dtype=self.np_dtype)
        self._ck(self._lib.pbvi_belief_update(self._h, a.ctypes.data_as(C.POINTER(C.c_int32)),
                                            o.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(out)))
        return out

    def belief_walk(self, b0: np.ndarray, actions, observations, restart=None):
        """Chained Bayes updates on the device (``pbvi_belief_walk``): returns ``(values [n,S] f64, first store id)``;
        belief ``i`` of the walk is already in the belief store under id ``first + i``."""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        o = np.ascontiguousarray(observations, dtype=np.int32)
        n = a.shape[0]
        if o.shape != (n,) or n == 0:
            raise ValueError('actions / observations must be equally long and non-empty')
        start = np.ascontiguousarray(b0, dtype=np.float64)
        if start.shape != (self.S,):
            raise ValueError(f'b0 must be [{self.S}]')
        rp = None
        if restart is not None:
            r = np.ascontiguousarray(restart, dtype=np.uint8)
            if r.shape != (n,):
                raise ValueError('restart must be [n]')
            rp = r.ctypes.data_as(C.POINTER(C.c_uint8))
        out = self._arena.empty((n, self.S), np.float64)           # kept by the caller as Belief values
        f64p, i32p = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        first = int(self._lib.pbvi_belief_walk(self._h, start.ctypes.data_as(f64p), n, a.ctypes.data_as(i32p),
                                               o.ctypes.data_as(i32p), rp, out.ctypes.data_as(f64p)))
        if first < 0:
            self._ck(first)
        return out, first

    def belief_walk_keys(self, n: int) -> np.ndarray:
        """``[n]`` uint64: position-weighted bit-pattern hashes of the fp64 rows of the last ``belief_walk`` (what ``_RowKey`` computes on
        the host), from the device."""
        keys = np.empty(n, dtype=np.uint64)
        self._ck(self._lib.pbvi_belief_walk_keys(self._h, n, keys.ctypes.data_as(C.POINTER(C.c_uint64))))
        return keys

    def belief_tag(self):
        """Tag that marks an object as resident in this engine's belief store (see ``row_ids``)."""
        return (self.serial, 'belief', self._store_epoch['belief'])

    def advance_beliefs(self, actions, observations, keep=None) -> int:
        """Simulator step on the resident block (``src/pomdp.py:3305-3329``): Bayes-update every belief with its
        ``(action, observation)`` and keep only the rows with ``keep[b]`` true, in order.  Returns the new B."""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        o = np.ascontiguousarray(observations, dtype=np.int32)
        if a.shape != (self.B,) or o.shape != (self.B,):
            raise ValueError('actions / observations must be [B]')
        kp = None
        if keep is not None:
            k = np.ascontiguousarray(keep, dtype=np.uint8)
            if k.shape != (self.B,):
                raise ValueError('keep must be [B]')
            kp = k.ctypes.data_as(C.POINTER(C.c_uint8))
        nb = C.c_int64(0)
        self._resident['belief'] = None
        self._ck(self._lib.pbvi_beliefs_advance(self._h, a.ctypes.data_as(C.POINTER(C.c_int32)),
                                              o.ctypes.data_as(C.POINTER(C.c_int32)), kp, C.byref(nb)))
        self.B = int(nb.value)
        return self.B

    def fetch_beliefs(self) -> np.ndarray:
        """The resident belief block, ``[B,S]`` in caller order."""
        out = np.empty((self.B, self.S), dtype=self.np_dtype)
        self._ck(self._lib.pbvi_beliefs_fetch(self._h, _ptr(out)))
        return out

    def set_formulation(self, which: str = 'auto') -> None:
        """Operand projected through the model: ``'auto'``, ``'alpha'`` (Gamma, the reference's order) or
        ``'belief'`` (beliefs pushed through every (a, o); cheaper when B << V)."""
        self._ck(self._lib.pbvi_set_formulation(self._h, {'auto': 0, 'alpha': 1, 'belief': 2}[which]))
        self._formulation = which

    @property
    def formulation(self) -> str:
        """The setting ``set_formulation`` last made (initially ``PBVI_FORMULATION`` or ``'auto'``, as the engine reads it)."""
        if self._formulation is None:
            env = os.environ.get('PBVI_FORMULATION', '')
            self._formulation = {'alpha': 'alpha', '1': 'alpha', 'belief': 'belief', '2': 'belief'}.get(env, 'auto')
        return self._formulation

    def set_fused_projection(self, enable=True) -> None:
        """fp32 scoring: Gamma tiles generated inside the score GEMM (``True``: where that is faster, i.e. R = 1; default) or
        projected first (``False``); ``2`` also fuses R = 2..7 (slower; tests).  Same scores bit for bit
        (``pbvi_set_fused_projection``)."""
        self._ck(self._lib.pbvi_set_fused_projection(self._h, int(enable)))

    def set_f64_screen(self, mode: str = 'auto') -> None:
        """fp64 engines: ``'off'`` (pure fp64 arithmetic), ``'auto'`` (fp32 screen + fp64 re-decision of near-ties when the
        score GEMM is large; default) or ``'always'`` (``pbvi_set_f64_screen``)."""
        self._ck(self._lib.pbvi_set_f64_screen(self._h, {'off': 0, 'auto': 1, 'always': 2}[mode]))

    def set_tie_window(self, rel: float) -> None:
        self._ck(self._lib.pbvi_set_tie_window(self._h, float(rel)))

    @property
    def device_bytes(self) -> int:
        return int(self._lib.pbvi_device_bytes(self._h))
