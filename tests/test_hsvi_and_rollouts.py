"""API rows beside the backup that the reference's notebooks call: HSVI (sawtooth upper bound + greedy descent),
``PBVI_Solver.test_n_simulations``, ``Model.get_coords / save / load_from_file``, ``SolverHistory.explored_beliefs``.
Expected values come from the reference itself (``tests/golden/make_golden.py hsvi`` -> ``hsvi_and_rollouts.npz``)."""
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN, load_npz
from pomdp_pbvi_exploration_amd import (Belief, BeliefValueMapping, HSVI_Solver, PBVI_Solver, ValueFunction,
                                        load_POMDP_file, synth)
from test_policy_eval import mirror_model

MODELS = os.path.join(GOLDEN, 'models')
FILES = {'tiger': 'tiger.95.POMDP', 'grid4x3': '4x3.95-no_loop_2_grid.POMDP'}


def z():
    return load_npz('hsvi_and_rollouts.npz')


def hsvi_file_solve(key, use_gpu=False):
    g = z()
    model, pbvi = load_POMDP_file(os.path.join(MODELS, FILES[key]))
    solver = HSVI_Solver(gamma=pbvi.gamma, eps=1e-6)
    np.random.seed(0)
    random.seed(0)
    exps, growth = (int(x) for x in g[f'{key}_cfg'])
    with np.errstate(divide='ignore', invalid='ignore'):
        vf, hist = solver.solve(model, expansions=exps, max_belief_growth=growth, use_gpu=use_gpu, print_progress=False,
                                history_tracking_level=2)
    return g, model, solver, vf, hist


def olfactory(R, on_gpu=False, dtype='f64'):
    m = synth.olfactory_model(H=15, W=40, R=R, f32=False)
    model = mirror_model(m)
    return m, (model.to_gpu(dtype) if on_gpu else model)


@pytest.mark.parametrize('key', ['tiger', 'grid4x3'])
def test_hsvi_solve_matches_reference_host(key):
    g, model, solver, vf, hist = hsvi_file_solve(key)
    assert hist.beliefs_counts == list(g[f'{key}_beliefs'])
    assert hist.alpha_vector_counts == list(g[f'{key}_alphas'])
    np.testing.assert_allclose(hist.value_function_changes, g[f'{key}_changes'], rtol=1e-12, atol=1e-12)
    assert np.array_equal(np.asarray(vf.actions), g[f'{key}_actions'])
    np.testing.assert_allclose(vf.alpha_vector_array, g[f'{key}_alpha'], rtol=1e-12, atol=1e-12)
    # the upper bound the descent built: same points, same values, same sawtooth interpolation at probe beliefs
    ub = solver._upper_bound
    np.testing.assert_allclose(np.array([b.values for b in ub.beliefs]), g[f'{key}_ub_points'], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(list(ub.belief_value_mapping.values()), g[f'{key}_ub_values'], rtol=1e-12)
    ub.update()
    got = [ub.evaluate(Belief(model, p)) for p in g[f'{key}_ub_probes']]
    np.testing.assert_allclose(got, g[f'{key}_ub_probe_values'], rtol=1e-12)
    assert len(hist.explored_beliefs) == hist.beliefs_counts[-1] and hist.solution is hist.value_functions[-1]


@pytest.mark.parametrize('R', [1, 5])
def test_hsvi_solve_olfactory_matches_reference_host(R):
    g = z()
    m, model = olfactory(R)
    np.random.seed(0)
    random.seed(0)
    exps, growth = (int(x) for x in g[f'olf{R}_cfg'])
    vf, hist = HSVI_Solver(gamma=m.gamma, eps=1e-6).solve(model, expansions=exps, max_belief_growth=growth, print_progress=False)
    assert hist.beliefs_counts == list(g[f'olf{R}_beliefs']) and hist.alpha_vector_counts == list(g[f'olf{R}_alphas'])
    assert np.array_equal(np.asarray(vf.actions), g[f'olf{R}_actions'])
    np.testing.assert_allclose(vf.alpha_vector_array, g[f'olf{R}_alpha'], rtol=1e-12, atol=1e-12)


def test_sawtooth_upper_bound_basics():
    model, _ = load_POMDP_file(os.path.join(MODELS, FILES['tiger']))
    corner = ValueFunction(model, np.array([[10.0, 2.0], [1.0, 8.0]]), [0, 1])
    ub = BeliefValueMapping(model, corner)
    b = Belief(model, np.array([0.5, 0.5]))
    assert ub.evaluate(b) == 9.0                                   # no points: the corner interpolation
    ub.add(b, 5.0)
    ub.add(Belief(model, np.array([0.5, 0.5])), 99.0)              # an existing point keeps its first value
    assert len(ub.beliefs) == 1 and ub.evaluate(b) == 5.0
    q = Belief(model, np.array([0.75, 0.25]))
    assert ub.evaluate(q) == pytest.approx(9.5 + (5.0 - 9.0) * 0.5)   # v0 + (v - corner.b) * min(q / b)


def run_tns(R, on_gpu=False, dtype='f64'):
    g = z()
    m, model = olfactory(R, on_gpu, dtype)
    vf = ValueFunction(model, g[f'olf{R}_alpha'], g[f'olf{R}_actions'].astype(int))
    n, horizon, seed = (int(x) for x in g[f'olf{R}_tns_cfg'])
    np.random.seed(seed)
    random.seed(seed)
    return g, PBVI_Solver(gamma=m.gamma).test_n_simulations(model, vf, n=n, horizon=horizon)


def check_tns(R, g, got):
    starts, done_at, rewards, disc = got
    assert np.array_equal(starts, g[f'olf{R}_tns_starts'])
    assert np.array_equal(done_at, g[f'olf{R}_tns_done_at'])
    assert np.array_equal(np.asarray(rewards, dtype=np.float64), g[f'olf{R}_tns_rewards'])
    np.testing.assert_allclose(np.asarray(disc, dtype=np.float64), g[f'olf{R}_tns_discounted'], rtol=1e-15)


@pytest.mark.parametrize('R', [1, 5])
def test_test_n_simulations_matches_reference_host(R):
    g, got = run_tns(R)
    check_tns(R, g, got)


def test_model_coords_and_pickle_round_trip(tmp_path):
    g = z()
    m, model = olfactory(1)
    assert np.array_equal(np.array(model.get_coords([0, 41, 599])), g['olf1_coords'])
    assert list(model.get_coords(41)) == list(g['olf1_coords'][1])
    model.save('olf', path=str(tmp_path / 'Models'))
    back = type(model).load_from_file(str(tmp_path / 'Models' / 'olf.pck'))
    assert not back.is_on_gpu and back._alt_model is None
    for attr in ('reachable_states', 'reachable_probabilities', 'reachable_transitional_observation_table',
                 'expected_rewards_table', 'observation_table', 'start_probabilities', 'state_grid'):
        assert np.array_equal(getattr(back, attr), getattr(model, attr)), attr
    assert back.end_states == model.end_states and back.state_count == model.state_count


# --------------------------------------------------------------------------- #
# with the HIP engine (backup, compute_change, belief block of the rollouts on the device)
# --------------------------------------------------------------------------- #
@pytest.mark.gpu
@pytest.mark.parametrize('key', ['tiger', 'grid4x3'])
def test_hsvi_solve_matches_reference_gpu(key):
    g, model, solver, vf, hist = hsvi_file_solve(key, use_gpu=True)
    assert hist.beliefs_counts == list(g[f'{key}_beliefs'])
    assert hist.alpha_vector_counts == list(g[f'{key}_alphas'])
    np.testing.assert_allclose(hist.value_function_changes, g[f'{key}_changes'], rtol=1e-9, atol=1e-9)
    assert np.array_equal(np.asarray(vf.actions), g[f'{key}_actions'])
    np.testing.assert_allclose(vf.alpha_vector_array, g[f'{key}_alpha'], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize('R', [1, 5])
@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_test_n_simulations_matches_reference_gpu(R, dtype):
    g, got = run_tns(R, on_gpu=True, dtype=dtype)
    check_tns(R, g, got)


@pytest.mark.gpu
def test_pickled_model_drops_the_engine(tmp_path):
    m, model = olfactory(1, on_gpu=True)
    model.save('gpu_side', path=str(tmp_path))
    back = type(model).load_from_file(str(tmp_path / 'gpu_side.pck'))
    assert not back.is_on_gpu and back._engine is None
    assert back.gpu_model.is_on_gpu                        # a fresh engine is built on demand
