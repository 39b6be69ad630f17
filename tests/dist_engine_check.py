"""Child ranks of test_sharded_backup_on_the_engine_two_ranks (tests/test_gpu_parity.py): launched with
``python -m torch.distributed.run --nproc-per-node 2``; both ranks share GPU 0, so the collective runs over gloo (RCCL
refuses two ranks on one device) with host carrier buffers -- everything else is the product path: one HIP engine per
rank, ``PBVI_Solver.backup`` / ``solve`` taking the sharded route by themselves."""
import os
import random
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
    from pomdp_pbvi_exploration_amd import (Belief, BeliefSet, FSVI_Solver, Model, PBVI_Solver, ValueFunction,
                                            load_POMDP_file, set_quiet, synth)
    from pomdp_pbvi_exploration_amd import dist as pdist
    set_quiet(True)
    pdist.enable(True)                                    # sharding is opt-in
    assert pdist.active() and world == 2

    # 1. the reference's seeded FSVI run of the 4x3 grid with the f64 engine, every backup sharded
    z = load_npz('grid4x3_fsvi.npz')
    model, _ = load_POMDP_file(os.path.join(REPO, 'tests', 'golden', 'models', '4x3.95-no_loop_2_grid.POMDP'))
    model.end_states = [3, 6]
    np.random.seed(0)
    random.seed(0)
    vf, hist = FSVI_Solver(gamma=0.95, eps=1e-6).solve(model, expansions=10, max_belief_growth=10, print_progress=False,
                                                       use_gpu=True)
    assert hist.alpha_vector_counts == list(z['alpha_counts']), (rank, hist.alpha_vector_counts)
    last = int(z['n_calls']) - 1
    np.testing.assert_allclose(vf.alpha_vector_array, z[f'c{last}_out_alpha'], rtol=1e-9, atol=1e-12)
    assert np.array_equal(vf.actions, z[f'c{last}_out_actions'])

    # 2. direct backups on a small olfactory model, both engine types, B odd (ragged shards) and B = 1 (empty shard):
    #    two consecutive steps (append on: the second one selects the rows the first one appended on every replica)
    #    equal the single-process engine's
    m = synth.olfactory_model(H=15, W=40, R=1, f32=True)
    alpha, acts = synth.alpha_set(m, 40)
    for dtype, rtol in (('f64', 1e-12), ('f32', 1e-6)):
        om = Model(states=m.S, actions=m.A, observations=m.O, reachable_states=m.reachable_states,
                   observation_table=m.observation_table, end_states=[m.goal], start_probabilities=list(m.start_belief))
        gm = om.to_gpu(dtype)
        solver = PBVI_Solver(gamma=m.gamma)
        for n_b, prune in ((37, False), (37, True), (1, False)):
            rows = synth.belief_points(m, n_b, max_depth=16)
            outs = {}
            for mode in ('single', 'sharded'):
                if mode == 'single':
                    os.environ['PBVI_NO_SHARD'] = '1'
                else:
                    os.environ.pop('PBVI_NO_SHARD', None)
                bs = BeliefSet(gm, [Belief(gm, r) for r in rows])
                vf0 = ValueFunction(gm, alpha, acts)
                v1 = solver.backup(gm, bs, vf0, append=True, belief_dominance_prune=prune)
                v2 = solver.backup(gm, bs, v1, append=True, belief_dominance_prune=prune)
                outs[mode] = (v1, v2)
            os.environ.pop('PBVI_NO_SHARD', None)
            for a, b in zip(outs['single'], outs['sharded']):
                assert len(a) == len(b), (dtype, n_b, prune, len(a), len(b))
                assert np.array_equal(a.actions, b.actions)
                np.testing.assert_allclose(np.asarray(b.alpha_vector_array, dtype=np.float64),
                                           np.asarray(a.alpha_vector_array, dtype=np.float64), rtol=rtol, atol=0)
            # the replicas' stores stayed in step: same number of rows on both ranks
            n_rows = torch.tensor([int(gm.engine._lib.pbvi_alpha_store_count(gm.engine._h))])
            both = [torch.zeros_like(n_rows) for _ in range(world)]
            dist.all_gather(both, n_rows)
            assert int(both[0]) == int(both[1]), [int(x) for x in both]
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print('sharded engine backup ok')


if __name__ == '__main__':
    main()
