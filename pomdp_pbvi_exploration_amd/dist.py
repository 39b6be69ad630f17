"""Belief-sharded backup across the GPUs of one node.

One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on the
GPU box, ``gloo`` in the CPU tests).  Every rank holds the whole model and alpha set;
the B beliefs are split into contiguous blocks of ceil(B/G) rows; after the local
backup ONE all-gather moves the new alpha rows ``[B/G, S]`` (+ actions, keep mask) so
every rank ends with the full ``[B, S]`` result in belief order and the replicas stay
identical.  No other collective is on the data path.  The reference has no
counterpart (single GPU, ``cupy.cuda.runtime.setDevice``).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous block ``[lo, hi)`` of rank ``rank`` and the common padded block size."""
    per = -(-n // world)
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi, per


class ShardedBackup:
    """Runs a backup sharded over the ranks of ``group`` and all-gathers the results.

    ``local_backup(beliefs_local) -> (alpha_new[b,S], actions[b], keep[b])`` is the
    per-rank work: on the GPU box it is the HIP engine writing straight into torch
    CUDA tensors (``EngineShard``); the CPU tests pass the host NumPy path.
    """

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def gather_rows(self, local_rows, local_actions, local_keep, n_total: int):
        """All-gather per-rank blocks (padded to ceil(B/G) rows) and trim to ``n_total``."""
        import torch
        lo, hi, per = shard_bounds(n_total, self.world, self.rank)
        S = local_rows.shape[1]
        dev = local_rows.device

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
        return all_rows[:n_total], all_acts[:n_total], all_keep[:n_total]

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


def gather_unique(dist, group, rows, count: int, index, actions, keep, n_total: int):
    """All-gather of deduplicated per-rank results.  Each rank contributes ``count`` unique alpha' rows
    (``rows[:count]``), its per-belief ``index`` into them, ``actions`` and ``keep``; blocks are padded to
    the largest count; two ``all_gather_into_tensor`` calls in all (the per-belief integers and the row counts
    travel together, then the rows).  Returns the concatenated
    unique rows ``[sum U_r, S]``, the global index ``[B]`` (offset per rank), actions and keep in belief
    order."""
    import torch
    world = dist.get_world_size(group)
    dev = rows.device
    S = rows.shape[1]
    per = index.shape[0]
    # exchange 1: everything that is one int per belief, plus this rank's row count, in one message
    meta = torch.empty(1 + 3 * per, dtype=torch.int32, device=dev)
    meta[0] = count
    meta[1:1 + per] = index
    meta[1 + per:1 + 2 * per] = actions
    meta[1 + 2 * per:] = keep
    flat = torch.empty(world * (1 + 3 * per), dtype=torch.int32, device=dev)     # gloo wants a flat output
    dist.all_gather_into_tensor(flat, meta, group=group)
    all_meta = flat.view(world, 1 + 3 * per)
    counts_h = all_meta[:, 0].tolist()
    # exchange 2: the unique rows, padded to the largest count so one all-gather suffices
    umax = max(max(counts_h), 1)
    if count == umax:
        send = rows[:umax].contiguous()
    else:
        send = torch.zeros((umax, S), dtype=rows.dtype, device=dev)
        send[:count] = rows[:count]
    all_rows = torch.empty((world * umax, S), dtype=rows.dtype, device=dev)
    dist.all_gather_into_tensor(all_rows, send, group=group)
    uniq = torch.cat([all_rows[r * umax: r * umax + counts_h[r]] for r in range(world)], dim=0)
    counts_d = all_meta[:, 0].to(torch.int64)                  # offsets on the device: no host-to-device copy in the step
    gidx = all_meta[:, 1:1 + per].to(torch.int64) + (torch.cumsum(counts_d, 0) - counts_d)[:, None]
    all_act = all_meta[:, 1 + per:1 + 2 * per].reshape(-1).to(actions.dtype)
    all_keep = all_meta[:, 1 + 2 * per:].reshape(-1).to(keep.dtype)
    return uniq, gidx.reshape(-1)[:n_total], all_act[:n_total], all_keep[:n_total]


def pack_exchange(keys, count: int, index, actions, keep):
    """The per-rank message of ``gather_packed`` from separate tensors (the engine writes it in one kernel,
    ``pbvi_backup_fetch_exchange``): ``[U | index[B] | actions[B] | keep[B] | keys[B][1+O] (first U valid)]`` int32."""
    import torch
    per = index.shape[0]
    body = torch.zeros((per, keys.shape[1]), dtype=torch.int32, device=index.device)
    body[:count] = keys[:count]
    head = torch.tensor([count], dtype=torch.int32, device=index.device)
    return torch.cat([head, index.to(torch.int32), actions.to(torch.int32), keep.to(torch.int32), body.reshape(-1)])


def gather_packed(dist, group, meta, per: int, key_width: int, n_total: int, assemble):
    """The exchange without rows.  A rank's alpha' rows are functions of their keys ``(a*, v*[a*, :])`` and of the
    replicated alpha set and model, so ONE ``all_gather_into_tensor`` of integers suffices: per rank the packed
    message of ``pack_exchange``.  Every rank then rebuilds all rows with
    ``assemble(all_keys [sum U_r, 1+O]) -> [sum U_r, S]`` (``pbvi_assemble_rows``: byte-identical to the rows the
    producing rank holds).  At C4 that is 28 KB per rank on the wire instead of 8 MB.
    Returns ``(unique rows [sum U_r, S], global index [n_total], actions [n_total], keep [n_total])``."""
    import torch
    world = dist.get_world_size(group)
    dev = meta.device
    n_meta = meta.shape[0]
    assert n_meta == 1 + 3 * per + per * key_width
    flat = torch.empty(world * n_meta, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(flat, meta, group=group)
    all_meta = flat.view(world, n_meta)
    counts_h = all_meta[:, 0].tolist()
    k0 = 1 + 3 * per
    all_keys = torch.cat([all_meta[r, k0:k0 + counts_h[r] * key_width].view(counts_h[r], key_width) for r in range(world)], dim=0)
    uniq = assemble(all_keys.contiguous())
    counts_d = all_meta[:, 0].to(torch.int64)                  # offsets on the device: no host-to-device copy in the step
    gidx = all_meta[:, 1:1 + per].to(torch.int64) + (torch.cumsum(counts_d, 0) - counts_d)[:, None]
    all_act = all_meta[:, 1 + per:1 + 2 * per].reshape(-1)
    all_keep = all_meta[:, 1 + 2 * per:1 + 3 * per].reshape(-1).to(torch.uint8)
    return uniq, gidx.reshape(-1)[:n_total], all_act[:n_total], all_keep[:n_total]


def gather_keys(dist, group, keys, count: int, index, actions, keep, n_total: int, assemble):
    """``gather_packed`` for callers that hold the pieces separately."""
    return gather_packed(dist, group, pack_exchange(keys, count, index, actions, keep), index.shape[0], keys.shape[1],
                         n_total, assemble)


class EngineShard:
    """Per-rank adapter: HIP engine results copied device-to-device into torch CUDA
    tensors that RCCL can send (torch is only the carrier of device memory here)."""

    def __init__(self, engine, gamma: float, belief_dominance_prune: bool = False):
        import torch
        self.torch = torch
        self.engine = engine
        self.gamma = gamma
        self.prune = belief_dominance_prune
        self.device = torch.device('cuda', engine.device)
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

    def run_resident_packed(self):
        """For ``gather_packed``: the engine packs count, index, actions, keep and the keys of its distinct rows into one
        device int32 buffer (``pbvi_backup_fetch_exchange``); no alpha' row leaves the engine.  Returns
        ``(meta, B, 1+O, stats)``."""
        t = self.torch
        stats = self.engine.run(self.gamma, self.prune)
        B, kw = self.engine.B, 1 + self.engine.O
        n = 1 + 3 * B + B * kw
        if self._keys is None or self._keys.shape[0] != n:
            self._keys = t.empty(n, dtype=t.int32, device=self.device)
        self.engine.fetch_exchange_into(self._keys.data_ptr())
        return self._keys, B, kw, stats

    def assemble(self, keys):
        """Rows for a device tensor of keys (device in, device out)."""
        t = self.torch
        dt = t.float32 if self.engine.dtype == 'f32' else t.float64
        out = t.empty((keys.shape[0], self.engine.S), dtype=dt, device=self.device)
        # `keys` was produced on torch's stream; the engine reads it on its own (non-blocking) stream
        t.cuda.current_stream(self.device).synchronize()
        self.engine.assemble_rows_into(keys.data_ptr(), keys.shape[0], self.gamma, out.data_ptr())
        return out

    def __call__(self, beliefs_local: np.ndarray):
        self.engine.set_beliefs(beliefs_local)
        rows, acts, keep, _ = self.run_resident()
        return rows, acts, keep
