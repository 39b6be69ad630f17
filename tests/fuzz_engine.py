"""Random models through the engine against the oracle, by hand on a GPU box (the suite runs 60 seeds of the same generator:
test_gpu_parity.py::test_random_models_against_the_oracle):

    PYTHONPATH=. python tests/fuzz_engine.py 0 1000            # seeds 0..999, both engine types
    PBVI_FORMULATION=belief PYTHONPATH=. python tests/fuzz_engine.py 2000 2300

Prints every case whose indices (other than between exact duplicates of an alpha row), actions or rows differ from the
oracle's.  Test infrastructure: uses oracle/."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import pbvi_oracle as orc                                   # noqa: E402
from pomdp_pbvi_exploration_amd.engine import Engine                    # noqa: E402
from test_gpu_parity import _random_case                                # noqa: E402


def run(seed, dtype):
    S, A, O, R, rs, rto, er, alpha, b, gamma = _random_case(seed)
    if dtype == 'f32':
        alpha, b = alpha.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)
        rto, er = rto.astype(np.float32).astype(np.float64), er.astype(np.float32).astype(np.float64)
    want_rows, want_a, want_v = orc.backup_core(alpha, b, rs, rto, er, gamma)
    eng = Engine(S, A, O, R, rs, rto, er, dtype=dtype)
    cast = (lambda x: x) if dtype == 'f64' else (lambda x: x.astype(np.float32))
    res = eng.backup_full(cast(alpha), cast(b), gamma)
    eng.close()
    diff = np.argwhere(res.best_alpha_ind != want_v)
    real = [tuple(i) for i in diff if not np.array_equal(alpha[res.best_alpha_ind[tuple(i)]], alpha[want_v[tuple(i)]])]
    tol = 1e-6 if dtype == 'f32' else 1e-12
    ok_rows = np.allclose(res.alpha, want_rows, rtol=tol, atol=tol * (np.abs(want_rows).max() + 1e-30))
    return real, len(diff) - len(real), np.array_equal(res.actions, want_a), ok_rows, (S, A, O, R, alpha.shape[0], b.shape[0])


def main():
    n0, n1 = int(sys.argv[1]), int(sys.argv[2])
    dtypes = sys.argv[3:] or ['f32', 'f64']
    bad = dup = 0
    t0 = time.time()
    for seed in range(n0, n1):
        for dtype in dtypes:
            real, n_dup, ok_a, ok_r, shape = run(seed, dtype)
            dup += n_dup
            if real or not ok_a or not ok_r:
                bad += 1
                print(f'MISMATCH seed {seed} {dtype} (S, A, O, R, V, B) = {shape}: {len(real)} indices, actions {ok_a}, rows {ok_r}', flush=True)
        if (seed - n0) % 50 == 49:
            print(f'... seed {seed}: {bad} mismatches, {dup} indices differing between exact duplicates, {time.time() - t0:.0f} s', flush=True)
    print(f'{n1 - n0} seeds x {len(dtypes)} engine types: {bad} mismatches ({dup} indices differed between exact duplicates of an alpha row)')


if __name__ == '__main__':
    main()
