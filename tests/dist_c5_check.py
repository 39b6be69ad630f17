"""Child ranks of test_c5_sharded_over_two_ranks_against_reference (tests/test_gpu_parity.py): BASELINE config 5's
workload -- |S|=30000, V=1024, B=8192 -- sharded over TWO ranks (one HIP engine each on this box's GPU; gloo carries
the exchange because RCCL refuses two ranks on one device; on an 8-GPU node the same code runs over RCCL with 8 ranks of
1024 beliefs).  Each rank generates only its own block of the global belief sequence, runs the product step
(dist.sharded_engine_step: local backup, one all-gather of integers, global dedup, append of the distinct rows to the
alpha store) and checks the merged result against the reference's outputs for all 8192 beliefs."""
import os
import sys

import numpy as np
import torch                                    # noqa: F401  torch first: its HIP runtime has to be the one that opens the device
import torch.distributed as dist

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    from conftest import load_npz
    from oracle import pbvi_oracle as orc                    # checker only
    from pomdp_pbvi_exploration_amd import synth
    from pomdp_pbvi_exploration_amd.dist import EngineShard, shard_bounds, sharded_engine_step
    from pomdp_pbvi_exploration_amd.engine import Engine
    z = load_npz('olfactory_c5_B8192.npz')
    B, V = int(z['B']), int(z['V'])
    lo, hi, per = shard_bounds(B, world, rank)
    m = synth.olfactory_model(R=1)
    alpha, _ = synth.alpha_set(m, V)
    beliefs = synth.belief_points(m, hi - lo, start=lo)
    eng = Engine(m.S, m.A, m.O, m.R, m.reachable_states, m.rto, m.expected_rewards, dtype='f32', device=0)
    eng.set_alpha(alpha)
    eng.set_beliefs(beliefs)
    shard = EngineShard(eng, m.gamma, carrier=torch.device('cpu'))
    n0 = int(eng._lib.pbvi_alpha_store_count(eng._h))
    first, n_rows, idx, act, keep, st = sharded_engine_step(shard, dist, None, B)
    assert idx.shape == (B,) and act.shape == (B,) and bool(keep.all())
    assert np.array_equal(act, z['core_actions'].astype(np.int64)), int(np.sum(act != z['core_actions']))
    # this rank's own block, index by index, against the reference
    res = eng.fetch()
    assert np.array_equal(res.best_alpha_ind, z['core_best'][lo:hi].astype(np.int64))
    # the globally distinct rows are in the store under consecutive ids on every rank; their byte-dedup is the
    # reference's ValueFunction over all 8192 beliefs
    assert first == n0 and int(eng._lib.pbvi_alpha_store_count(eng._h)) == n0 + n_rows
    # the rows themselves, for the value checks: the exchange once more, rows to the host this time
    from pomdp_pbvi_exploration_amd.dist import exchange_keys
    meta, _, kw, _ = shard.run_resident_packed(per)
    keys, idx2, act2, _ = exchange_keys(dist, None, meta, per, kw, B)
    assert np.array_equal(idx2, idx) and np.array_equal(act2, act) and keys.shape[0] == n_rows
    rows = eng.assemble_rows(keys, m.gamma)
    full_sum = rows.astype(np.float64).sum(axis=1)[idx]
    np.testing.assert_allclose(full_sum, z['row_sum'], rtol=1e-6)
    np.testing.assert_allclose(rows.astype(np.float64)[idx[z['sample_b']], z['sample_s']], z['sample_val'], rtol=1e-6, atol=1e-12)
    assert len(orc.dedup_rows(rows[idx], act)[1]) == int(z['n_unique'])
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(f'c5 sharded ok: {B} beliefs over {world} ranks, {n_rows} distinct keys, {int(z["n_unique"])} distinct rows, '
              f'local backup {st["ms_total"]:.2f} ms')


if __name__ == '__main__':
    main()
