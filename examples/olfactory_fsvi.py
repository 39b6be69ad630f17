"""End-to-end FSVI solve of the synthetic olfactory model on the HIP engine.

Reproduces the protocol of the reference's headline run (Olfactory_Alternation_Paper_Wrap.ipynb:
FSVI, gamma 0.99, N expansions of <=100 beliefs, `solver.solve(model, expansions, max_belief_growth=100,
use_gpu=True)`) through this repo's `src`-compatible API and prints the same SolverHistory summary
figures (mean backup time, |V|).  BASELINE.md: the reference's CuPy run reports 0.6306 s mean backup,
its NumPy run 9.9306 s (S=22021, 300 expansions, unnamed hardware).

    python examples/olfactory_fsvi.py --expansions 60 --dtype f32
"""
import argparse
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pomdp_pbvi_exploration_amd import FSVI_Solver, Model, set_quiet, synth   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--expansions', type=int, default=60)
    ap.add_argument('--growth', type=int, default=100)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f64'])
    ap.add_argument('--grid', default='75x400')
    ap.add_argument('--cpu', action='store_true', help='run the host NumPy path instead (slow)')
    args = ap.parse_args()
    set_quiet(True)
    H, W = (int(x) for x in args.grid.split('x'))
    m = synth.olfactory_model(H=H, W=W, R=1, f32=False)     # exact fp64 tables (observation rows sum to 1)
    model = Model(states=m.S, actions=m.A, observations=m.O, reachable_states=m.reachable_states,
                  observation_table=m.observation_table, end_states=[m.goal], start_probabilities=list(m.start_belief))
    np.random.seed(0)
    random.seed(0)
    solver = FSVI_Solver(gamma=m.gamma, eps=1e-6)
    t0 = time.perf_counter()
    vf, hist = solver.solve(model, expansions=args.expansions, max_belief_growth=args.growth, use_gpu=not args.cpu,
                            engine_dtype=args.dtype, print_progress=False)
    wall = time.perf_counter() - t0
    bt = np.array(hist.backup_times)
    print(hist.summary)
    print(f'S={m.S} expansions={len(hist.expansion_times)} |V|={len(vf)} |B|={hist.beliefs_counts[-1]} '
          f'wall={wall:.2f}s backup mean={bt.mean() * 1e3:.2f} ms median={np.median(bt) * 1e3:.2f} ms '
          f'last={bt[-1] * 1e3:.2f} ms expand mean={np.mean(hist.expansion_times) * 1e3:.1f} ms')


if __name__ == '__main__':
    main()
