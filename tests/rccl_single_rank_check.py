"""Child process of test_rccl_exchange_single_rank_matches_direct_fetch (tests/test_gpu_parity.py)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    from conftest import load_npz
    from pomdp_pbvi_exploration_amd.engine import Engine
    from pomdp_pbvi_exploration_amd.dist import EngineShard, gather_packed, gather_unique
    z = load_npz('olfactory_small_R5.npz')
    rs, rto, er = z['reachable_states'].astype(np.int64), z['rto'].astype(np.float64), z['expected_rewards'].astype(np.float64)
    gamma = float(z['gamma'])
    eng = Engine(rto.shape[0], rto.shape[1], rto.shape[2], rto.shape[3], rs, rto, er, dtype='f32')
    eng.set_alpha(z['alpha'])
    eng.set_beliefs(z['beliefs'])
    shard = EngineShard(eng, gamma)
    meta, per, kw, _ = shard.run_resident_packed()
    uniq, gidx, acts, keep = gather_packed(dist, None, meta, per, kw, per, shard.assemble)
    res = eng.fetch()
    assert np.array_equal(uniq.cpu().numpy(), res.unique_alpha)
    assert np.array_equal(gidx, res.index)
    assert np.array_equal(acts, res.actions)
    assert np.array_equal(keep, res.keep.astype(bool))
    # the product path's step: exchange + global dedup + append to the alpha store, rows never leave the device
    from pomdp_pbvi_exploration_amd.dist import sharded_engine_step
    n0 = eng._lib.pbvi_alpha_store_count(eng._h)
    first, n_rows, i3, a3, k3, _ = sharded_engine_step(shard, dist, None, per)
    assert first == n0 and n_rows == res.unique_alpha.shape[0] and eng._lib.pbvi_alpha_store_count(eng._h) == n0 + n_rows
    assert np.array_equal(i3, res.index) and np.array_equal(a3, res.actions)
    alpha_before = z['alpha']
    eng.select_alpha(np.arange(first, first + n_rows))        # the appended rows are the backup's rows
    eng.set_beliefs(z['beliefs'][:4])
    val, _ = eng.max_value_resident()
    exp = (z['beliefs'][:4].astype(np.float64) @ res.unique_alpha.astype(np.float64).T).max(axis=1)
    np.testing.assert_allclose(val, exp, rtol=1e-12)
    eng.set_alpha(alpha_before)
    eng.set_beliefs(z['beliefs'])
    # the row exchange (PBVI_EXCHANGE=rows) gives the same rows
    rows, count, idx, a2, k2, _ = shard.run_resident_unique()
    u2, g2, _, _ = gather_unique(dist, None, rows, count, idx, a2, k2, per)
    assert np.array_equal(u2.cpu().numpy(), res.unique_alpha) and np.array_equal(g2.cpu().numpy(), res.index)
    dist.destroy_process_group()
    eng.close()
    print('rccl single-rank exchange ok')


if __name__ == '__main__':
    main()
