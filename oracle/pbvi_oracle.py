"""CPU oracle for the PBVI alpha-vector backup path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the reference algorithm
(PimLb/POMDP_PBVI_Exploration @ 2024_08_07) for the one hot path this repo
accelerates.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker /
the timed CPU baseline.  Nothing under ``pomdp_pbvi_exploration_amd/`` imports
it: the product path is the HIP engine behind ``include/pbvi_hip.h``.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
itself (``/root/reference/src/pomdp.py``) in the build container and stores its
inputs/outputs as fixtures; ``tests/test_oracle.py`` checks every function here
against those fixtures and against the known answers the reference's notebooks
hold.

Every function cites the reference file:line it restates.  All arithmetic is
float64 with int64 indices, exactly as the reference (``src/mdp.py:335``).
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------- #
# Model tables
# --------------------------------------------------------------------------- #
def reachable_from_dense(transition_table: np.ndarray):
    """Padded-ELL reachable states / probabilities from a dense T[S,A,S].

    Restates ``src/mdp.py:306-335`` (scan ``T[s,a,:] > 0``; pad short lists with
    the lowest state indices not already in the list) and ``src/mdp.py:348``
    (probabilities gathered from the table, so padding entries carry the
    table's value at the pad state, i.e. 0).
    """
    S, A, _ = transition_table.shape
    lists = [[np.flatnonzero(transition_table[s, a] > 0).tolist() for a in range(A)] for s in range(S)]
    R = max(len(l) for row in lists for l in row)
    for s in range(S):
        for a in range(A):
            l = lists[s][a]
            cand = 0
            while len(l) < R:
                if cand not in l:
                    l.append(cand)
                cand += 1
    rs = np.array(lists, dtype=np.int64)
    rp = transition_table[np.arange(S)[:, None, None], np.arange(A)[None, :, None], rs]
    return rs, rp


def rto_table(reach_states: np.ndarray, reach_probs: np.ndarray, observation_table: np.ndarray) -> np.ndarray:
    """RTO[s,a,o,r] = P(r|s,a) * O[rs[s,a,r], a, o]   (``src/pomdp.py:201-202``)."""
    S, A, R = reach_states.shape
    O = observation_table.shape[2]
    reach_obs = observation_table[reach_states[:, :, None, :],
                                  np.arange(A)[None, :, None, None],
                                  np.arange(O)[None, None, :, None]]
    return np.einsum('sar,saor->saor', reach_probs, reach_obs)


def expected_rewards(rto: np.ndarray, reach_states: np.ndarray, reward_table: np.ndarray) -> np.ndarray:
    """ER[s,a] = sum_{o,r} RTO[s,a,o,r] * rewards[s,a,rs[s,a,r],o]   (``src/pomdp.py:237,251``)."""
    S, A, O, R = rto.shape
    reach_rewards = reward_table[np.arange(S)[:, None, None, None],
                                 np.arange(A)[None, :, None, None],
                                 reach_states[:, :, :, None],
                                 np.arange(O)[None, None, None, :]]
    return np.einsum('saor,saro->sa', rto, reach_rewards)


# --------------------------------------------------------------------------- #
# Belief update (input generator for the path)
# --------------------------------------------------------------------------- #
def belief_update(belief: np.ndarray, a: int, o: int, reach_states: np.ndarray, rto: np.ndarray) -> np.ndarray:
    """Bayes step b' ~ sum_s b[s] RTO[s,a,o,r] scattered to rs[s,a,r]   (``src/pomdp.py:405-411``)."""
    S = belief.shape[0]
    w = rto[:, a, o, :] * belief[:, None]
    nb = np.bincount(reach_states[:, a, :].flatten(), weights=w.flatten(), minlength=S)
    nb /= np.sum(nb)
    return nb


# --------------------------------------------------------------------------- #
# The backup
# --------------------------------------------------------------------------- #
def gamma_projection(alpha: np.ndarray, reach_states: np.ndarray, rto: np.ndarray, gamma: float) -> np.ndarray:
    """Gamma[a,o,v,s] = gamma * sum_r RTO[s,a,o,r] * alpha[v, rs[s,a,r]]   (``src/pomdp.py:1485-1491``)."""
    V = alpha.shape[0]
    alpha_r = alpha[np.arange(V)[:, None, None, None], reach_states[None, :, :, :]]      # V,S,A,R
    return gamma * np.einsum('saor,vsar->aovs', rto, alpha_r)


def backup_core(alpha: np.ndarray, beliefs: np.ndarray, reach_states: np.ndarray, rto: np.ndarray,
                exp_rewards: np.ndarray, gamma: float):
    """One point-based backup before pruning / dedup.

    Restates ``src/pomdp.py:1485-1506`` statement by statement.  Returns
    ``(alpha_new[B,S], best_action[B], best_alpha_ind[B,A,O])``.
    """
    A = rto.shape[1]
    O = rto.shape[2]
    S = rto.shape[0]
    gamma_a_o_t = gamma_projection(alpha, reach_states, rto, gamma)                        # :1489
    best_alpha_ind = np.argmax(np.tensordot(beliefs, gamma_a_o_t, (1, 3)), axis=3)           # :1495
    best_alphas_per_o = gamma_a_o_t[np.arange(A)[None, :, None, None],
                                    np.arange(O)[None, None, :, None],
                                    best_alpha_ind[:, :, :, None],
                                    np.arange(S)[None, None, None, :]]                       # :1497
    alpha_a = exp_rewards.T + np.sum(best_alphas_per_o, axis=2)                              # :1502
    best_actions = np.argmax(np.einsum('bas,bs->ba', alpha_a, beliefs), axis=1)              # :1505
    alpha_new = np.take_along_axis(alpha_a, best_actions[:, None, None], axis=1)[:, 0, :]    # :1506
    return alpha_new, best_actions, best_alpha_ind


def belief_dominance_mask(alpha_old: np.ndarray, beliefs: np.ndarray, alpha_new: np.ndarray) -> np.ndarray:
    """keep[b] = b.alpha_new[b] > max_v b.alpha_old[v]   (``src/pomdp.py:1510-1512``)."""
    best_value_per_belief = np.sum(beliefs * alpha_new, axis=1)
    old_best_value_per_belief = np.max(np.matmul(beliefs, alpha_old.T), axis=1)
    return best_value_per_belief > old_best_value_per_belief


def dedup_rows(values: np.ndarray, actions: np.ndarray):
    """ValueFunction constructor dedup: exact-byte key, FIRST position, LAST action.

    Restates ``src/mdp.py:660-669`` (a dict keyed on ``values.tobytes()``
    overwritten in order keeps the first insertion position and the last
    AlphaVector object, hence the last duplicate's action).
    """
    d = {}
    for row, act in zip(values, actions):
        d[row.tobytes()] = (row, int(act))
    if not d:
        return values[:0], np.asarray(actions[:0], dtype=np.int64)
    rows = np.array([v[0] for v in d.values()])
    acts = np.array([v[1] for v in d.values()], dtype=np.int64)
    return rows, acts


def extend_rows(new_values, new_actions, old_values, old_actions):
    """``ValueFunction.extend``: new vectors first, then old; on a byte-identical
    pair the OLD vector (and its action) is kept at the NEW position
    (``src/mdp.py:773-774``)."""
    d = {}
    for row, act in zip(new_values, new_actions):
        d[row.tobytes()] = (row, int(act))
    for row, act in zip(old_values, old_actions):
        d[row.tobytes()] = (row, int(act))
    rows = np.array([v[0] for v in d.values()])
    acts = np.array([v[1] for v in d.values()], dtype=np.int64)
    return rows, acts


def backup(alpha: np.ndarray, alpha_actions: np.ndarray, beliefs: np.ndarray, reach_states: np.ndarray,
           rto: np.ndarray, exp_rewards: np.ndarray, gamma: float,
           append: bool = False, belief_dominance_prune: bool = True):
    """Full ``PBVI_Solver.backup`` (``src/pomdp.py:1447-1524``) on plain arrays.

    ``alpha`` / ``alpha_actions`` are the (already deduplicated) input value
    function; returns the deduplicated output ``(alpha_out[V',S], actions[V'])``.
    """
    alpha_new, best_actions, _ = backup_core(alpha, beliefs, reach_states, rto, exp_rewards, gamma)
    if belief_dominance_prune:
        keep = belief_dominance_mask(alpha, beliefs, alpha_new)
        alpha_new = alpha_new[keep]
        best_actions = best_actions[keep]
    rows, acts = dedup_rows(alpha_new, best_actions)
    if append:
        if len(rows) == 0:
            return alpha.copy(), np.asarray(alpha_actions, dtype=np.int64).copy()
        rows, acts = extend_rows(rows, acts, alpha, alpha_actions)
    return rows, acts


# --------------------------------------------------------------------------- #
# Domination prune
# --------------------------------------------------------------------------- #
def prune_dominated_mask(alpha: np.ndarray) -> np.ndarray:
    """keep[i] iff exactly one row j (itself) has alpha[j] >= alpha[i] in every state.

    Restates ``ValueFunction.prune(level=2)``, ``src/mdp.py:857-866``.
    """
    keep = np.zeros(alpha.shape[0], dtype=bool)
    for i, v in enumerate(alpha):
        is_dom_by = np.all(alpha >= v, axis=1)
        keep[i] = (np.count_nonzero(is_dom_by) == 1)
    return keep


# --------------------------------------------------------------------------- #
# compute_change (the caller-side GEMM + row-max)
# --------------------------------------------------------------------------- #
def max_value_per_belief(alpha: np.ndarray, beliefs: np.ndarray) -> np.ndarray:
    """max_v b.alpha_v   (``src/pomdp.py:2165``)."""
    return np.max(np.matmul(beliefs, alpha.T), axis=1)


def compute_change(alpha_a: np.ndarray, alpha_b: np.ndarray, beliefs: np.ndarray) -> float:
    """``PBVI_Solver.compute_change`` (``src/pomdp.py:2165-2167``)."""
    return float(np.max(np.abs(max_value_per_belief(alpha_b, beliefs) - max_value_per_belief(alpha_a, beliefs))))


# --------------------------------------------------------------------------- #
# Memory-bounded variant for the S~30k configs (same results, tiled over V)
# --------------------------------------------------------------------------- #
def backup_core_tiled(alpha, beliefs, reach_states, rto, exp_rewards, gamma, v_tile: int = 128):
    """``backup_core`` tiled over V so Gamma[A,O,V,S] never exists whole.

    Same statements as ``src/pomdp.py:1485-1506`` applied per V-tile with a running
    first-max argmax (``np.argmax`` keeps the lowest index on ties, so a strict
    ``>`` update across ascending tiles reproduces it).  Used by the tests and the
    CPU baseline at |S|~30k where the untiled form needs ~14 GB.
    """
    B, S = beliefs.shape
    _, A, O, R = rto.shape
    V = alpha.shape[0]
    best_val = np.full((B, A, O), -np.inf)
    best_ind = np.zeros((B, A, O), dtype=np.int64)
    for v0 in range(0, V, v_tile):
        g = gamma_projection(alpha[v0:v0 + v_tile], reach_states, rto, gamma)
        sc = np.tensordot(beliefs, g, (1, 3))                     # B,A,O,v
        loc = np.argmax(sc, axis=3)
        val = np.take_along_axis(sc, loc[..., None], axis=3)[..., 0]
        upd = val > best_val
        best_val = np.where(upd, val, best_val)
        best_ind = np.where(upd, loc + v0, best_ind)
    # alpha_a[b,a,s] = ER[s,a] + sum_o Gamma[a,o,v*[b,a,o],s], recomputed from alpha rows
    alpha_a = np.empty((B, A, S))
    sidx = np.arange(S)
    for a in range(A):
        acc = np.zeros((B, S))
        for o in range(O):
            rows = alpha[best_ind[:, a, o]]                        # B,S
            g = np.zeros((B, S))
            for r in range(R):
                g += rto[sidx, a, o, r][None, :] * rows[:, reach_states[:, a, r]]
            acc += gamma * g
        alpha_a[:, a, :] = exp_rewards[:, a][None, :] + acc
    best_actions = np.argmax(np.einsum('bas,bs->ba', alpha_a, beliefs), axis=1)
    alpha_new = np.take_along_axis(alpha_a, best_actions[:, None, None], axis=1)[:, 0, :]
    return alpha_new, best_actions, best_ind
