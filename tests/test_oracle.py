"""The CPU oracle against fixtures produced by the reference itself (tests/golden/make_golden.py)
and against the known answers stored in the reference's notebooks."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_npz
from oracle import pbvi_oracle as orc

RT = dict(rtol=1e-12, atol=1e-13)   # same NumPy statements; only the BLAS build may differ between hosts


def _tables(z):
    return (z['reachable_states'].astype(np.int64), z['rto'].astype(np.float64), z['expected_rewards'].astype(np.float64))


@pytest.mark.parametrize('R', [1, 5])
def test_backup_matches_reference_small(R):
    z = load_npz(f'olfactory_small_R{R}.npz')
    rs, rto, er = _tables(z)
    alpha, acts = z['alpha'].astype(np.float64), z['alpha_actions'].astype(np.int64)
    b = z['beliefs'].astype(np.float64)
    g = float(z['gamma'])
    for tag, append, prune in (('plain', False, False), ('prune', False, True), ('append', True, False)):
        rows, a = orc.backup(alpha, acts, b, rs, rto, er, g, append=append, belief_dominance_prune=prune)
        assert rows.shape == z[f'{tag}_alpha'].shape
        np.testing.assert_allclose(rows, z[f'{tag}_alpha'], **RT)
        assert np.array_equal(a, z[f'{tag}_actions'])
    new, a_star, v_star = orc.backup_core(alpha, b, rs, rto, er, g)
    assert np.array_equal(a_star, z['core_actions']) and np.array_equal(v_star, z['core_best'])
    np.testing.assert_allclose(new, z['core_alpha'], **RT)
    t_new, t_a, t_v = orc.backup_core_tiled(alpha, b, rs, rto, er, g, v_tile=16)
    assert np.array_equal(t_a, a_star) and np.array_equal(t_v, v_star)
    np.testing.assert_allclose(t_new, new, **RT)


def test_backup_matches_reference_grid4x3_all_calls():
    z = load_npz('grid4x3_fsvi.npz')
    t = load_npz('grid4x3_tables.npz')
    rs, rto, er = t['reachable_states'], t['rto'], t['expected_rewards']
    n = int(z['n_calls'])
    assert n == 10
    assert list(z['alpha_counts']) == [4, 10, 17, 26, 32, 38, 45, 50, 59, 67, 76]
    for i in range(n):
        alpha, acts, b = z[f'c{i}_alpha'], z[f'c{i}_actions'], z[f'c{i}_beliefs']
        rows, a = orc.backup(alpha, acts, b, rs, rto, er, 0.95, append=bool(z[f'c{i}_append']), belief_dominance_prune=False)
        assert rows.shape == z[f'c{i}_out_alpha'].shape
        np.testing.assert_allclose(rows, z[f'c{i}_out_alpha'], **RT)
        assert np.array_equal(a, z[f'c{i}_out_actions'])
        rows, a = orc.backup(alpha, acts, b, rs, rto, er, 0.95, append=False, belief_dominance_prune=True)
        assert rows.shape == z[f'c{i}_prune_alpha'].shape
        np.testing.assert_allclose(rows, z[f'c{i}_prune_alpha'], **RT)
        assert np.array_equal(a, z[f'c{i}_prune_actions'])


def test_tables_from_dense_match_reference():
    for name in ('tiger_tables.npz', 'grid4x3_tables.npz'):
        t = load_npz(name)
        S, A, R = t['reachable_states'].shape
        # rebuild the dense table the reference derived the ELL form from
        T = np.zeros((S, A, S))
        for s in range(S):
            for a in range(A):
                for r in range(R):
                    T[s, a, t['reachable_states'][s, a, r]] += t['reachable_probabilities'][s, a, r]
        rs, rp = orc.reachable_from_dense(T)
        assert np.array_equal(rs, t['reachable_states'])
        assert np.array_equal(rp, t['reachable_probabilities'])
        assert np.array_equal(orc.rto_table(rs, rp, t['observation_table']), t['rto'])


def test_kat1_tiger_one_step():
    t = load_npz('tiger_tables.npz')
    rs, rto, er = t['reachable_states'], t['rto'], t['expected_rewards']
    alpha0, acts0 = orc.dedup_rows(er.T.copy(), np.arange(3))
    rows, a = orc.backup(alpha0, acts0, np.array([[0.5, 0.5]]), rs, rto, er, 0.95, belief_dominance_prune=False)
    np.testing.assert_allclose(rows, [[-1.95, -1.95]], rtol=1e-15)      # -1 + 0.95 * (-1), hand-derived
    assert list(a) == [0]
    kat = json.load(open(os.path.join(GOLDEN, 'kat.json')))
    assert kat['kat1_alpha'] == rows.tolist() and kat['kat1_actions'] == [0]


def test_kat2_belief_update():
    t = load_npz('tiger_tables.npz')
    nb = orc.belief_update(np.array([0.5, 0.5]), 0, 0, t['reachable_states'], t['rto'])
    np.testing.assert_allclose(nb, [0.85, 0.15], rtol=1e-15)            # tiger_problem_from_file.ipynb:272


def test_prune_dominated_semantics():
    a = np.array([[1., 2.], [0., 1.], [2., 0.], [1., 2.]])
    # rows 0 and 3 are identical: each is >= the other, so BOTH go (reference loop, src/mdp.py:860-863)
    assert orc.prune_dominated_mask(a).tolist() == [False, False, True, False]
    assert orc.prune_dominated_mask(a[:3]).tolist() == [True, False, True]


def test_prune_level2_matches_the_reference_fixture():
    """prune_level2.npz: what the reference's ValueFunction.prune(2) kept (make_golden.py prune) on a seeded set with
    dominated, dominating and one-state-apart rows and a +0 / -0 pair."""
    z = load_npz('prune_level2.npz')
    mask = orc.prune_dominated_mask(z['alpha'])
    assert np.array_equal(np.flatnonzero(mask), z['kept'])
    assert not mask[-1] and not mask[-2]                    # the pair that differs only in the sign of a zero


def test_dedup_first_position_last_action():
    v = np.array([[1., 2.], [3., 4.], [1., 2.]])
    rows, acts = orc.dedup_rows(v, np.array([0, 1, 2]))
    assert rows.tolist() == [[1., 2.], [3., 4.]] and acts.tolist() == [2, 1]
    rows, acts = orc.extend_rows(np.array([[9., 9.], [3., 4.]]), [5, 6], v[:2], [0, 1])
    assert rows.tolist() == [[9., 9.], [3., 4.], [1., 2.]] and acts.tolist() == [5, 1, 0]


@pytest.mark.parametrize('R', [1, 5, '5_1024'])
def test_full_size_summary_consistent(R):
    """The |S|=30000 fixtures hold only output summaries; here they are checked for internal
    consistency (the heavy comparison against them is the GPU test)."""
    path = os.path.join(GOLDEN, f'olfactory_full_R{R}.npz')
    if not os.path.exists(path):
        pytest.skip('full-size fixture not generated')
    z = np.load(path, allow_pickle=False)
    B = int(z['B'])
    assert z['core_actions'].shape == (B,) and z['core_best'].shape == (B, 6, 3)
    assert z['core_best'].min() >= 0 and z['core_best'].max() < int(z['V'])
    assert np.all(np.isfinite(z['row_sum'])) and np.all(np.isfinite(z['b_dot']))
