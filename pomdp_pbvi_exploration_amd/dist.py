"""Belief-sharded backup across the GPUs of one node.

One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in the CPU
tests).  Every rank holds the whole model and alpha set; the B beliefs are split into contiguous blocks of
``per = ceil(B/G)`` rows; after the local backup ONE all-gather makes every rank hold the whole result, the ranks
deduplicate it globally and every replica appends the same rows to its alpha store, so the replicas stay identical
from one backup to the next.  No other collective is on the data path.  The reference has no counterpart (single
GPU, ``cupy.cuda.runtime.setDevice``, ``Experiments/Olfactory Navigation/run_test.py:12``).

Two exchanges:

* **keys** (engine path, default): the alpha' row of a belief is a function of its key ``(a*, v*[a*, :])`` and of the
  replicated alpha set and model only, so a rank contributes integers -- per-belief index / action / keep and the
  keys of its distinct rows, one message of ``1 + 3 per + per (1+O)`` int32 (``pbvi_backup_fetch_exchange_padded``) --
  and every rank rebuilds the globally distinct rows against its own replica (``pbvi_assemble_rows_store``: straight
  into the alpha store, device to device).
* **rows** (host NumPy path, and ``PBVI_EXCHANGE=rows`` for A/B): padded row blocks.

Every message is padded to the common block size, so ragged splits (B % G != 0) and empty shards (B < G) send
equally long messages; a rank without beliefs contributes a zero-count message instead of skipping the collective.

``sharded_backup`` is what ``PBVI_Solver.backup`` calls when sharding was ASKED FOR (``PBVI_Solver.shard_beliefs =
True``, ``enable(True)`` or ``PBVI_SHARD=1``) and a process group with more than one rank is up: same arguments, same
return value as the single-process backup.  It is opt-in because it is only correct when every rank calls it with
the same model, belief set and value function -- ranks that run independent experiments (the reference's
one-process-per-GPU ``run_test.py`` pattern) must not be coupled by a collective they did not ask for.  What can be
checked is checked: every message carries ``(n_total, |V|, a fingerprint of the alpha set's store ids)`` behind the
engine's payload and every rank compares all of them after the gather (``ReplicaMismatch``).

Hardware status: the >= 2-rank RCCL path has not run on GPUs in any round (one-GPU boxes); it is covered by gloo tests
with one engine per rank and by a one-rank RCCL group.
"""
from __future__ import annotations

import os
import sys

import numpy as np


_ENABLED = None          # None: the environment decides (PBVI_SHARD=1); True / False: enable() was called


class ReplicaMismatch(RuntimeError):
    """The ranks of a sharded backup do not hold the same problem (belief count, alpha set)."""


def enable(on: bool = True) -> None:
    """Process-wide opt-in (or opt-out) for belief sharding in ``PBVI_Solver.backup``."""
    global _ENABLED
    _ENABLED = bool(on)


def requested(solver=None) -> bool:
    """Was sharding asked for?  The solver's ``shard_beliefs`` attribute wins, then ``enable()``, then ``PBVI_SHARD=1``."""
    want = getattr(solver, 'shard_beliefs', None)
    if want is None:
        want = _ENABLED
    if want is None:
        want = os.environ.get('PBVI_SHARD', '') not in ('', '0')
    return bool(want) and not os.environ.get('PBVI_NO_SHARD')


def active(group=None, solver=None) -> bool:
    """True when sharding was asked for (``requested``) and this process is one rank of a multi-rank
    ``torch.distributed`` job."""
    if 'torch' not in sys.modules or not requested(solver):
        return False
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


# Trailer every rank appends to its exchange message: what must be equal on all ranks for the keys to mean the same rows
TRAILER = 4
_MAGIC = 0x50425649


def alpha_fingerprint(engine) -> int:
    """31-bit fingerprint of the working alpha set's identity: the store ids it was selected by (replicas that appended
    the same rows in the same order hold the same ids), or just its size when it was uploaded as an array."""
    import zlib
    ids = engine._resident.get('alpha') if hasattr(engine, '_resident') else None
    if ids is None:
        return int(engine.alpha_count) & 0x7fffffff
    return zlib.crc32(np.ascontiguousarray(ids, dtype=np.int32).tobytes()) & 0x7fffffff


def trailer_values(n_total: int, V: int, fingerprint: int) -> np.ndarray:
    return np.array([_MAGIC, n_total & 0x7fffffff, V & 0x7fffffff, fingerprint & 0x7fffffff], dtype=np.int32)


def check_trailers(all_meta: np.ndarray, n_meta: int, mine: np.ndarray) -> None:
    """all_meta [world, n_meta + TRAILER]: every rank's trailer must equal this rank's."""
    got = all_meta[:, n_meta:n_meta + TRAILER]
    bad = np.flatnonzero(np.any(got != mine[None, :], axis=1))
    if bad.size:
        r = int(bad[0])
        raise ReplicaMismatch(f'sharded backup: rank {r} sent (magic, n_total, |V|, alpha fingerprint) = {got[r].tolist()}, '
                              f'this rank has {mine.tolist()}: the ranks do not hold the same belief set / value function '
                              f'(belief sharding needs replicated inputs; unset shard_beliefs / PBVI_SHARD for independent runs)')


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous block ``[lo, hi)`` of rank ``rank`` and the common padded block size."""
    per = -(-n // world)
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi, per


def _valid_mask(n_total: int, world: int, per: int) -> np.ndarray:
    """[world * per] bool: which slots of the per-padded concatenation are real beliefs."""
    m = np.zeros(world * per, dtype=bool)
    for r in range(world):
        lo, hi, _ = shard_bounds(n_total, world, r)
        m[r * per: r * per + (hi - lo)] = True
    return m


def _carrier_device(dist, group, device_index=None):
    """Where collective operands must live: CUDA for nccl (= RCCL) -- the engine's own device when there is one, else
    torch's current device -- and host memory for gloo."""
    import torch
    backend = str(dist.get_backend(group)).lower()
    if 'nccl' not in backend:
        return torch.device('cpu')
    return torch.device('cuda', torch.cuda.current_device() if device_index is None else int(device_index))


class ShardedBackup:
    """Row exchange: runs a backup sharded over the ranks of ``group`` and all-gathers per-belief rows.

    ``local_backup(beliefs_local) -> (alpha_new[b,S], actions[b], keep[b])`` is the per-rank work: the host NumPy
    statements in the CPU path, the HIP engine writing into torch CUDA tensors (``EngineShard``) for ``PBVI_EXCHANGE=rows``.
    """

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def gather_rows(self, local_rows, local_actions, local_keep, n_total: int):
        """All-gather per-rank blocks (padded to ceil(B/G) rows) and trim the padding of every block."""
        import torch
        lo, hi, per = shard_bounds(n_total, self.world, self.rank)
        S = local_rows.shape[1]
        dev = local_rows.device
        assert local_rows.shape[0] == hi - lo, f'rank {self.rank} holds {local_rows.shape[0]} rows, its shard has {hi - lo}'

        def padded(t, shape):
            if t.shape[0] == per:
                return t.contiguous()
            out = torch.zeros(shape, dtype=t.dtype, device=dev)
            out[: t.shape[0]] = t
            return out

        rows = padded(local_rows, (per, S))
        acts = padded(local_actions, (per,))
        keep = padded(local_keep, (per,))
        all_rows = torch.empty((self.world * per, S), dtype=rows.dtype, device=dev)
        all_acts = torch.empty((self.world * per,), dtype=acts.dtype, device=dev)
        all_keep = torch.empty((self.world * per,), dtype=keep.dtype, device=dev)
        self.dist.all_gather_into_tensor(all_rows, rows, group=self.group)
        self.dist.all_gather_into_tensor(all_acts, acts, group=self.group)
        self.dist.all_gather_into_tensor(all_keep, keep, group=self.group)
        if self.world * per == n_total:
            return all_rows, all_acts, all_keep
        valid = torch.from_numpy(_valid_mask(n_total, self.world, per)).to(dev)
        return all_rows[valid], all_acts[valid], all_keep[valid]

    def run(self, local_backup, beliefs_all: np.ndarray):
        """Shard ``beliefs_all`` [B,S] by rank, run the local backup, all-gather."""
        import torch
        n = beliefs_all.shape[0]
        lo, hi, _ = shard_bounds(n, self.world, self.rank)
        rows, acts, keep = local_backup(beliefs_all[lo:hi])
        if not torch.is_tensor(rows):
            rows = torch.from_numpy(np.ascontiguousarray(rows))
            acts = torch.from_numpy(np.ascontiguousarray(acts, dtype=np.int64))
            keep = torch.from_numpy(np.ascontiguousarray(keep, dtype=np.uint8))
        return self.gather_rows(rows, acts, keep, n)


def _check_common_per(per: int, n_total: int, world: int):
    want = -(-n_total // world)
    if per != want:
        raise ValueError(f'exchange block size {per} is not ceil({n_total}/{world}) = {want}: every rank must pad its '
                         f'message to the common block size')


def gather_unique(dist, group, rows, count: int, index, actions, keep, n_total: int, per: int = None):
    """Row exchange of deduplicated per-rank results (``PBVI_EXCHANGE=rows``).  Each rank contributes ``count`` unique
    alpha' rows (``rows[:count]``), its per-belief ``index`` into them, ``actions`` and ``keep`` (its own shard's
    length; padded here to the common ``per = ceil(n_total / world)``); blocks of rows are padded to the largest
    count; two ``all_gather_into_tensor`` calls in all.  Returns the concatenated unique rows ``[sum U_r, S]``, the
    global index ``[n_total]`` (offset per rank), actions and keep in belief order."""
    import torch
    world = dist.get_world_size(group)
    dev = rows.device
    S = rows.shape[1]
    if per is None:
        per = -(-n_total // world)
    _check_common_per(per, n_total, world)
    b_loc = index.shape[0]
    if b_loc > per:
        raise ValueError(f'this rank holds {b_loc} beliefs, more than the block size {per}')
    # exchange 1: everything that is one int per belief, plus this rank's row count, in one message
    meta = torch.zeros(1 + 3 * per, dtype=torch.int32, device=dev)
    meta[0] = count
    meta[1:1 + b_loc] = index
    meta[1 + per:1 + per + b_loc] = actions
    meta[1 + 2 * per:1 + 2 * per + b_loc] = keep
    flat = torch.empty(world * (1 + 3 * per), dtype=torch.int32, device=dev)     # gloo wants a flat output
    dist.all_gather_into_tensor(flat, meta, group=group)
    all_meta = flat.view(world, 1 + 3 * per)
    counts_h = all_meta[:, 0].tolist()
    # exchange 2: the unique rows, padded to the largest count so one all-gather suffices
    umax = max(max(counts_h), 1)
    if count == umax and rows.shape[0] >= umax:
        send = rows[:umax].contiguous()
    else:
        send = torch.zeros((umax, S), dtype=rows.dtype, device=dev)
        send[:count] = rows[:count]
    all_rows = torch.empty((world * umax, S), dtype=rows.dtype, device=dev)
    dist.all_gather_into_tensor(all_rows, send, group=group)
    uniq = torch.cat([all_rows[r * umax: r * umax + counts_h[r]] for r in range(world)], dim=0)
    counts_d = all_meta[:, 0].to(torch.int64)                  # offsets on the device: no host-to-device copy in the step
    gidx = (all_meta[:, 1:1 + per].to(torch.int64) + (torch.cumsum(counts_d, 0) - counts_d)[:, None]).reshape(-1)
    all_act = all_meta[:, 1 + per:1 + 2 * per].reshape(-1).to(actions.dtype)
    all_keep = all_meta[:, 1 + 2 * per:].reshape(-1).to(keep.dtype)
    if world * per != n_total:
        valid = torch.from_numpy(_valid_mask(n_total, world, per)).to(dev)
        gidx, all_act, all_keep = gidx[valid], all_act[valid], all_keep[valid]
    return uniq, gidx, all_act, all_keep


def pack_exchange(keys, count: int, index, actions, keep, per: int = None):
    """The per-rank message of the key exchange from separate tensors (the engine writes it in one kernel,
    ``pbvi_backup_fetch_exchange_padded``): ``[U | index[per] | actions[per] | keep[per] | keys[per][1+O] (first U
    valid)]`` int32, zeros behind this rank's beliefs."""
    import torch
    b_loc = index.shape[0]
    per = b_loc if per is None else per
    kw = keys.shape[1]
    meta = torch.zeros(1 + 3 * per + per * kw, dtype=torch.int32, device=index.device)
    meta[0] = count
    meta[1:1 + b_loc] = index.to(torch.int32)
    meta[1 + per:1 + per + b_loc] = actions.to(torch.int32)
    meta[1 + 2 * per:1 + 2 * per + b_loc] = keep.to(torch.int32)
    k0 = 1 + 3 * per
    meta[k0:k0 + count * kw] = keys[:count].reshape(-1).to(torch.int32)
    return meta


def merge_exchange(all_meta: np.ndarray, per: int, key_width: int, n_total: int):
    """Host side of the key exchange.  ``all_meta [world, >= 1 + 3 per + per kw]`` int32 (every rank's message, a
    trailer may follow the payload) -> ``(keys [n, kw], index [n_total], actions [n_total], keep [n_total])``: the
    globally distinct keys in order of first occurrence over ranks, and per belief (global belief order) the
    position of its key in that list.  Native (``pbvi_exchange_merge``): one pass with a hash table."""
    from .engine import load_library
    lib = load_library()
    all_meta = np.ascontiguousarray(all_meta, dtype=np.int32)
    world, stride = all_meta.shape
    total = int(np.clip(all_meta[:, 0], 0, per).sum())
    keys = np.empty((max(total, 1), key_width), dtype=np.int32)
    idx = np.empty(n_total, dtype=np.int32)
    act = np.empty(n_total, dtype=np.int32)
    keep = np.empty(n_total, dtype=np.uint8)
    n = int(lib.pbvi_exchange_merge(all_meta.ctypes.data, world, stride, per, key_width, n_total, keys.ctypes.data,
                                    idx.ctypes.data, act.ctypes.data, keep.ctypes.data))
    if n < 0:
        raise ValueError('corrupt exchange message: ' + (lib.pbvi_last_error() or b'').decode())
    return keys[:n], idx.astype(np.int64), act.astype(np.int64), keep.astype(bool)


def exchange_keys(dist, group, meta, per: int, key_width: int, n_total: int, trailer=None, timing=None):
    """ONE ``all_gather_into_tensor`` of the per-rank int32 messages (``meta``: a torch tensor on the backend's
    carrier device; ``1 + 3 per + per kw`` payload entries, ``+ TRAILER`` when ``trailer`` -- this rank's
    ``trailer_values`` -- is given: it is written behind the payload here and compared across ranks after the
    gather), then the host-side merge.  Returns ``merge_exchange``'s tuple.  ``timing``: a dict that receives the
    wall-clock split ``gather_ms`` / ``to_host_ms`` / ``merge_ms``."""
    import time
    import torch
    world = dist.get_world_size(group)
    _check_common_per(per, n_total, world)
    n_meta = 1 + 3 * per + per * key_width
    n_msg = int(meta.shape[0])
    if n_msg != n_meta + TRAILER and not (trailer is None and n_msg == n_meta):
        raise ValueError(f'exchange message has {n_msg} entries, block size {per} needs {n_meta} (+ {TRAILER} with a trailer)')
    if trailer is not None:
        meta[n_meta:] = torch.from_numpy(trailer).to(meta.device)
    t0 = time.perf_counter()
    flat = torch.empty(world * n_msg, dtype=torch.int32, device=meta.device)
    dist.all_gather_into_tensor(flat, meta, group=group)
    if timing is not None and flat.is_cuda:
        torch.cuda.synchronize(flat.device)
    t1 = time.perf_counter()
    host = flat.view(world, n_msg).cpu().numpy()
    t2 = time.perf_counter()
    if trailer is not None:
        check_trailers(host, n_meta, trailer)
    out = merge_exchange(host, per, key_width, n_total)
    if timing is not None:
        timing.update(gather_ms=(t1 - t0) * 1e3, to_host_ms=(t2 - t1) * 1e3, merge_ms=(time.perf_counter() - t2) * 1e3)
    return out


def gather_packed(dist, group, meta, per: int, key_width: int, n_total: int, assemble):
    """Key exchange + rows: ``assemble(keys [n, 1+O] int32 ndarray) -> rows [n, S]`` rebuilds the globally distinct
    rows (``pbvi_assemble_rows`` / ``_store``: byte-identical to the rows the producing rank holds).  At C4 that is
    28 KB per rank on the wire instead of 8 MB.  Returns ``(rows, index [n_total], actions, keep)``."""
    keys, idx, act, keep = exchange_keys(dist, group, meta, per, key_width, n_total)
    return assemble(keys), idx, act, keep


def gather_keys(dist, group, keys, count: int, index, actions, keep, n_total: int, assemble, per: int = None):
    """``gather_packed`` for callers that hold the pieces separately."""
    world = dist.get_world_size(group)
    per = -(-n_total // world) if per is None else per
    return gather_packed(dist, group, pack_exchange(keys, count, index, actions, keep, per), per, keys.shape[1],
                         n_total, assemble)


class EngineShard:
    """Per-rank adapter between the HIP engine and the collective: the engine writes its results into buffers the
    backend can send (torch is only the carrier: CUDA tensors for RCCL, host tensors for gloo)."""

    def __init__(self, engine, gamma: float, belief_dominance_prune: bool = False, carrier=None):
        import torch
        self.torch = torch
        self.engine = engine
        self.gamma = gamma
        self.prune = belief_dominance_prune
        self.device = carrier if carrier is not None else torch.device('cuda', engine.device)
        self._bufs = None
        self._keys = None

    def buffers(self, b: int):
        t = self.torch
        if self._bufs is None or self._bufs[0].shape[0] != b:
            dt = t.float32 if self.engine.dtype == 'f32' else t.float64
            self._bufs = (t.empty((b, self.engine.S), dtype=dt, device=self.device),
                          t.empty((b,), dtype=t.int32, device=self.device),
                          t.empty((b,), dtype=t.uint8, device=self.device),
                          t.empty((b,), dtype=t.int32, device=self.device))
        return self._bufs

    def run_resident(self):
        """Backup of the belief block already resident on this rank's engine; full per-belief rows."""
        stats = self.engine.run(self.gamma, self.prune)
        rows, acts, keep, _ = self.buffers(self.engine.B)
        self.engine.fetch_into(rows.data_ptr(), acts.data_ptr(), keep.data_ptr())
        return rows, acts, keep, stats

    def run_resident_unique(self):
        """Same, deduplicated: ``(rows[B,S] (first U valid), U, index[B], actions[B], keep[B], stats)``."""
        stats = self.engine.run(self.gamma, self.prune)
        rows, acts, keep, idx = self.buffers(self.engine.B)
        self.engine.fetch_into(0, acts.data_ptr(), keep.data_ptr())
        self.engine.fetch_unique_into(rows.data_ptr(), idx.data_ptr())
        return rows, self.engine.unique_count, idx, acts, keep, stats

    def message(self, per: int):
        """The int32 carrier tensor of one exchange message for block size ``per``: the engine's payload + the trailer."""
        t = self.torch
        n = self.engine.exchange_size(per) + TRAILER
        if self._keys is None or self._keys.shape[0] != n:
            self._keys = t.zeros(n, dtype=t.int32, device=self.device)
        return self._keys

    def trailer(self, n_total: int) -> np.ndarray:
        return trailer_values(n_total, int(self.engine.alpha_count), alpha_fingerprint(self.engine))

    def run_resident_packed(self, per: int = None):
        """For the key exchange: run, then the engine packs count, index, actions, keep and the keys of its distinct rows
        into one int32 buffer (``pbvi_backup_fetch_exchange_padded``); no alpha' row leaves the engine.  Returns
        ``(meta, per, 1+O, stats)``."""
        stats = self.engine.run(self.gamma, self.prune)
        per = self.engine.B if per is None else per
        meta = self.message(per)
        self.engine.fetch_exchange_into(meta.data_ptr(), per)
        return meta, per, 1 + self.engine.O, stats

    def empty_message(self, per: int):
        """What a rank without beliefs sends: a zero-count message of the common length."""
        meta = self.message(per)
        meta.zero_()
        return meta, per, 1 + self.engine.O

    def assemble(self, keys):
        """Rows for keys (ndarray or tensor) against the resident alpha set, as a tensor on the carrier device."""
        t = self.torch
        if t.is_tensor(keys):
            keys = keys.cpu().numpy()
        rows = self.engine.assemble_rows(np.ascontiguousarray(keys, dtype=np.int32), self.gamma)
        return t.from_numpy(rows).to(self.device)

    def __call__(self, beliefs_local: np.ndarray):
        self.engine.set_beliefs(beliefs_local)
        rows, acts, keep, _ = self.run_resident()
        return rows, acts, keep


def sharded_engine_step(shard: EngineShard, dist, group, n_total: int, store: bool = True, timing=None):
    """One sharded backup of the belief blocks RESIDENT on the ranks' engines (``bench.py --gpus N``): local backup,
    key exchange, global dedup, and every replica appends the globally distinct rows to its alpha store
    (``pbvi_assemble_rows_store``).  Returns ``(first store id, n distinct rows, index [n_total], actions, keep, stats)``.
    ``timing``: a dict that receives the host-side split of the step in ms (``backup_pack_ms``: local backup + the
    engine writing its message, ``gather_ms``, ``to_host_ms``, ``merge_ms``, ``append_ms``)."""
    import time
    world = dist.get_world_size(group)
    per = -(-n_total // world)
    eng = shard.engine
    t0 = time.perf_counter()
    if eng.B > 0:
        meta, per, kw, stats = shard.run_resident_packed(per)
    else:
        meta, per, kw = shard.empty_message(per)
        stats = {}
    t1 = time.perf_counter()
    keys, idx, act, keep = exchange_keys(dist, group, meta, per, kw, n_total, trailer=shard.trailer(n_total), timing=timing)
    t2 = time.perf_counter()
    first = -1
    if store and len(keys):
        _, first = eng.assemble_rows_store(keys, shard.gamma, want_rows=False)
    if timing is not None:
        timing.update(backup_pack_ms=(t1 - t0) * 1e3, append_ms=(time.perf_counter() - t2) * 1e3)
    return first, len(keys), idx, act, keep, stats


def _first_occurrence(idx: np.ndarray):
    """Distinct values of ``idx`` in order of first occurrence, and where each first occurs."""
    if idx.size == 0:
        return idx[:0], idx[:0]
    first = np.unique(idx, return_index=True)[1]
    first.sort()
    return idx[first], first


def sharded_backup(solver, model, belief_set, value_function, belief_dominance_prune: bool, group=None):
    """``PBVI_Solver.backup`` (``src/pomdp.py:1447-1519``, before the ``append`` union) with the beliefs sharded over
    the ranks of ``group``.  Every rank calls it with identical arguments (replicated model, belief set and value
    function) and gets the identical ``ValueFunction`` back: the single-process result."""
    import torch
    import torch.distributed as dist
    from .mdp import AlphaVector, ValueFunction
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    beliefs = belief_set.belief_list
    n_total = len(beliefs)
    lo, hi, per = shard_bounds(n_total, world, rank)
    carrier = _carrier_device(dist, group, value_function.model.engine.device if value_function.is_on_gpu else None)

    if not value_function.is_on_gpu:
        # host mirror: the reference's NumPy statements on this rank's block, per-belief rows exchanged
        S = model.state_count
        if hi > lo:
            rows, acts, keep = solver._backup_numpy(model, belief_set.belief_array[lo:hi], value_function.alpha_vector_array,
                                                    belief_dominance_prune, return_mask=True)
        else:
            rows, acts, keep = np.zeros((0, S)), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=bool)
        sb = ShardedBackup(group)
        t_rows, t_acts, t_keep = sb.gather_rows(torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float64)).to(carrier),
                                                torch.from_numpy(np.ascontiguousarray(acts, dtype=np.int64)).to(carrier),
                                                torch.from_numpy(np.ascontiguousarray(keep, dtype=np.uint8)).to(carrier), n_total)
        k = t_keep.cpu().numpy().astype(bool)
        return ValueFunction(model, t_rows.cpu().numpy()[k], t_acts.cpu().numpy()[k])

    eng = value_function.model.engine
    eng.sync_rows('alpha', value_function.alpha_vector_list, lambda v: v.values, owner=value_function)
    shard = EngineShard(eng, solver.gamma, belief_dominance_prune, carrier=carrier)
    if hi > lo:
        if hi - lo > 65535:
            raise NotImplementedError('sharded backup: at most 65535 beliefs per rank and call')
        eng.sync_rows('belief', beliefs[lo:hi], lambda b: b.values)
        meta, per, kw, _ = shard.run_resident_packed(per)
    else:
        meta, per, kw = shard.empty_message(per)
    keys, idx, act, keep = exchange_keys(dist, group, meta, per, kw, n_total, trailer=shard.trailer(n_total))
    if belief_dominance_prune:
        idx, act = idx[keep], act[keep]
    used, first_pos = _first_occurrence(idx)              # the order the reference's byte-dedup produces
    if used.size == 0:
        return ValueFunction(value_function.model, [])
    rows, first = eng.assemble_rows_store(keys[used], solver.gamma)
    tag = eng.store_tag('alpha')
    vectors = []
    for k, (row, a) in enumerate(zip(rows, act[first_pos])):
        v = AlphaVector(row, int(a))
        v._dev = (tag, first + k)
        vectors.append(v)
    return ValueFunction(value_function.model, vectors)
