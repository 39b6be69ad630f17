"""Belief-sharded backup over 2 ranks (gloo on CPU): the all-gather layer returns exactly the
single-process result, in belief order, for even and ragged splits."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, load_npz


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_beliefs, out_dir):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import pbvi_oracle as orc          # checker for the gathered result
        from pomdp_pbvi_exploration_amd.dist import ShardedBackup, shard_bounds
        z = np.load(os.path.join(REPO, 'tests', 'golden', 'olfactory_small_R5.npz'), allow_pickle=False)
        rs, rto, er = z['reachable_states'].astype(np.int64), z['rto'].astype(np.float64), z['expected_rewards'].astype(np.float64)
        alpha, b = z['alpha'].astype(np.float64), z['beliefs'].astype(np.float64)[:n_beliefs]
        g = float(z['gamma'])

        def local(beliefs_local):                         # host NumPy statements per shard
            if beliefs_local.shape[0] == 0:
                return np.zeros((0, b.shape[1])), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.uint8)
            new, act, _ = orc.backup_core(alpha, beliefs_local, rs, rto, er, g)
            return new, act, np.ones(len(act), dtype=np.uint8)

        sb = ShardedBackup()
        rows, acts, keep = sb.run(local, b)
        full_new, full_act, _ = orc.backup_core(alpha, b, rs, rto, er, g)
        assert rows.shape == (n_beliefs, b.shape[1])
        np.testing.assert_allclose(rows.numpy(), full_new, rtol=1e-13, atol=0)
        assert np.array_equal(acts.numpy(), full_act)
        assert keep.numpy().all()
        lo, hi, per = shard_bounds(n_beliefs, world, rank)
        assert per == -(-n_beliefs // world) and 0 <= lo <= hi <= n_beliefs
        open(os.path.join(out_dir, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_beliefs', [64, 37, 1])
def test_sharded_backup_world2(tmp_path, n_beliefs):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_beliefs, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')


def test_shard_bounds_cover_everything():
    from pomdp_pbvi_exploration_amd.dist import shard_bounds
    for n in (1, 7, 8, 1024, 8191):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
