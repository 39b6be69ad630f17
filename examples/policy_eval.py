"""Policy evaluation on the HIP engine: solve the synthetic olfactory model with FSVI, then run
`Agent.run_n_simulations_parallel` with the belief block resident on the GPU.

The reference publishes one timing for this step (sim_runtime_test.ipynb:223): 1000 simulations x
300 steps in 41.8 s on its CuPy path (S=22021, unnamed GPU).  This script runs the same call shape.

    python examples/policy_eval.py --expansions 60 --n 1000 --max-steps 300
"""
import argparse
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pomdp_pbvi_exploration_amd import FSVI_Solver, Model, set_quiet, synth   # noqa: E402
from pomdp_pbvi_exploration_amd.pomdp import Agent                            # noqa: E402
from pomdp_pbvi_exploration_amd.mdp import VI_Solver                         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--expansions', type=int, default=60)
    ap.add_argument('--growth', type=int, default=100)
    ap.add_argument('--n', type=int, default=1000)
    ap.add_argument('--max-steps', type=int, default=300)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f64'])
    ap.add_argument('--grid', default='75x400')
    ap.add_argument('--cpu-steps', type=int, default=0, help='also time this many steps of the host NumPy path')
    args = ap.parse_args()
    set_quiet(True)
    H, W = (int(x) for x in args.grid.split('x'))
    m = synth.olfactory_model(H=H, W=W, R=1, f32=False)
    model = Model(states=m.S, actions=m.A, observations=m.O, reachable_states=m.reachable_states,
                  observation_table=m.observation_table, end_states=[m.goal], start_probabilities=list(m.start_belief))
    # MDP value iteration that seeds FSVI: device sweeps vs the host NumPy loop (same result, see tests)
    t0 = time.perf_counter()
    mdp_dev, h_dev = VI_Solver(gamma=m.gamma, eps=1e-6).solve(model, use_gpu=True, print_progress=False)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    mdp_host, h_host = VI_Solver(gamma=m.gamma, eps=1e-6).solve(model, use_gpu=False, print_progress=False)
    t_host = time.perf_counter() - t0
    print(f'value iteration: {len(h_dev.iteration_times)} sweeps  device {t_dev:.3f}s  host NumPy {t_host:.3f}s  '
          f'identical rows: {np.array_equal(mdp_dev.alpha_vector_array, mdp_host.alpha_vector_array)}', flush=True)

    np.random.seed(0)
    random.seed(0)
    t0 = time.perf_counter()
    vf, hist = FSVI_Solver(gamma=m.gamma, eps=1e-6, mdp_policy=mdp_dev).solve(model, expansions=args.expansions, max_belief_growth=args.growth,
                                                          use_gpu=True, engine_dtype=args.dtype, print_progress=False)
    print(f'solve: S={m.S} expansions={len(hist.expansion_times)} |V|={len(vf)} in {time.perf_counter() - t0:.2f}s', flush=True)

    agent = Agent(vf.model, vf)
    np.random.seed(1)
    t0 = time.perf_counter()
    totals, hists = agent.run_n_simulations_parallel(n=args.n, max_steps=args.max_steps, print_progress=False,
                                                     print_stats=True)
    wall = time.perf_counter() - t0
    steps = sum(len(h.actions) for h in hists)
    print(f'gpu ({args.dtype}): n={args.n} max_steps={args.max_steps} |V|={len(vf)} wall={wall:.2f}s '
          f'belief-steps={steps} ({steps / wall:.0f} belief-steps/s)  reference CuPy: 41.8 s for 1000 x 300', flush=True)

    if args.cpu_steps > 0:
        host_agent = Agent(model, vf.to_cpu())
        np.random.seed(1)
        t0 = time.perf_counter()
        _, hh = host_agent.run_n_simulations_parallel(n=args.n, max_steps=args.cpu_steps, print_progress=False,
                                                      print_stats=False)
        wall_c = time.perf_counter() - t0
        steps_c = sum(len(h.actions) for h in hh)
        same = all(h.actions[:args.cpu_steps] == g.actions[:args.cpu_steps] for h, g in zip(hh, hists))
        print(f'host NumPy: {args.cpu_steps} steps wall={wall_c:.2f}s ({steps_c / wall_c:.0f} belief-steps/s); '
              f'same actions as the GPU run over those steps: {same}', flush=True)


if __name__ == '__main__':
    main()
