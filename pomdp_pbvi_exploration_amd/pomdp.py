"""Host-side mirror of the reference's ``src/pomdp.py`` around the backup path.

Same class names, signatures and container semantics as the reference so its
notebooks' ``from src.pomdp import *`` keeps working (``src/`` at the repo root
re-exports this module).  What differs is the seam the reference implements with
CuPy array-module dispatch (``src/pomdp.py:1482``, ``:2243-2264``): here
``use_gpu=True`` / ``.gpu_model`` / ``.to_gpu()`` bind the objects to a HIP
engine (``engine.Engine``) and ``PBVI_Solver.backup`` then runs on hand-written
gfx950 kernels through the C-ABI of ``include/pbvi_hip.h``.  With
``use_gpu=False`` the NumPy statements of the reference run on the host
(BASELINE config 0).  A GPU request never degrades to the CPU path: if the HIP
library is missing the call raises.
"""
from __future__ import annotations

import copy            # noqa: F401  (notebooks rely on these leaked names)
import random
from datetime import datetime
from typing import Tuple, Union

import numpy as np

from .mdp import AlphaVector, ValueFunction, VI_Solver, log, _log, set_quiet   # noqa: F401
from .mdp import Model as MDP_Model
from .mdp import Solver as MDP_Solver
from .mdp import RewardSet, _RowKey, _draw_index             # noqa: F401
from .mdp import SimulationHistory as MDP_SimulationHistory
from .mdp import Simulation as MDP_Simulation
from . import dist as _dist

gpu_support = True


class Model(MDP_Model):
    """POMDP model: MDP tables + observation table, fused ``RTO[s,a,o,r]`` and
    ``expected_rewards_table[s,a]`` (``src/pomdp.py:147-254``)."""

    def __init__(self, states, actions, observations, transitions=None, reachable_states=None, rewards=None,
                 observation_table=None, rewards_are_probabilistic: bool = False, state_grid=None,
                 start_probabilities=None, end_states: list = [], end_actions: list = []):
        super().__init__(states=states, actions=actions, transitions=transitions, reachable_states=reachable_states,
                         rewards=-1, rewards_are_probabilistic=rewards_are_probabilistic, state_grid=state_grid,
                         start_probabilities=start_probabilities, end_states=end_states, end_actions=end_actions)
        S, A = self.state_count, self.action_count
        self.observation_labels = [f'o_{i}' for i in range(observations)] if isinstance(observations, int) else observations
        self.observation_count = len(self.observation_labels)
        self.observations = np.arange(self.observation_count)
        O = self.observation_count

        if observation_table is None:
            rnd = np.random.rand(S, A, O)
            self.observation_table = rnd / np.sum(rnd, axis=2, keepdims=True)
        else:
            self.observation_table = np.array(observation_table)
            assert self.observation_table.shape == (S, A, O), \
                f"Observations table doesnt have the right shape, it should be SxAxO (expected: {(S, A, O)}, received: {self.observation_table.shape})."
        _log(f'POMDP model: {O} observations')

        rs = self.reachable_states
        reach_obs = self.observation_table[rs[:, :, None, :], self.actions[None, :, None, None], self.observations[None, None, :, None]]
        self.reachable_transitional_observation_table = np.einsum('sar,saor->saor', self.reachable_probabilities, reach_obs)

        self.immediate_reward_table = None
        self.immediate_reward_function = None
        if rewards is None:
            if len(self.end_states) > 0 or len(self.end_actions) > 0:
                self.immediate_reward_function = self._end_reward_function
            else:
                self.immediate_reward_table = np.random.rand(S, A, S, O)
        elif callable(rewards):
            self.immediate_reward_function = rewards
        else:
            self.immediate_reward_table = np.array(rewards)
            assert self.immediate_reward_table.shape == (S, A, S, O), \
                f"Rewards table doesnt have the right shape, it should be SxAxSxO (expected: {(S, A, S, O)}, received {self.immediate_reward_table.shape})"

        if self.immediate_reward_table is not None:
            reach_r = self.immediate_reward_table[self.states[:, None, None, None], self.actions[None, :, None, None],
                                                  rs[:, :, :, None], self.observations[None, None, None, :]]
        else:
            reach_r = np.fromfunction(lambda s, a, r, o: self.immediate_reward_function(
                s.astype(int), a.astype(int), rs[s.astype(int), a.astype(int), r.astype(int)], o.astype(int)), (*rs.shape, O))
        self._min_reward = float(np.min(reach_r))
        self._max_reward = float(np.max(reach_r))
        self.expected_rewards_table = np.einsum('saor,saro->sa', self.reachable_transitional_observation_table, reach_r)

    def _end_reward_function(self, s, a, sn, o):
        return (np.isin(sn, self.end_states) | np.isin(a, self.end_actions)).astype(int)

    def reward(self, s: int, a: int, s_p: int, o: int):
        """Reward of ``(s, a, s_p, o)``; a Bernoulli draw when rewards are probabilities (``src/pomdp.py:259-285``)."""
        if self.immediate_reward_table is not None:
            r = float(self.immediate_reward_table[s, a, s_p, o])
        else:
            r = float(self.immediate_reward_function(s, a, s_p, o))
        if self.rewards_are_probabilistic:
            return 1 if random.random() < r else 0
        return r

    def observe(self, s_p: int, a: int) -> int:
        return int(self.observations[_draw_index(self.observation_table[s_p, a])])


class Belief:
    """Probability distribution over states; Bayes update through the
    reachable-state tables (``src/pomdp.py:311-421``)."""

    def __new__(cls, *args, **kwargs):
        inst = super().__new__(cls)
        inst._bytes_repr = None
        inst._key = None
        inst._successors = {}
        return inst

    def __init__(self, model: Model, values: Union[np.ndarray, None] = None):
        assert model is not None
        self.model = model
        if values is not None:
            assert values.shape[0] == model.state_count, "Belief must contain be of dimension |S|"
            total = np.sum(values)
            assert np.round(total, decimals=3) == 1.0, f"States probabilities in belief must sum to 1 (found: {total})"
            self._values = values
        else:
            self._values = model.start_probabilities

    @property
    def values(self) -> np.ndarray:
        return self._values

    @property
    def bytes_repr(self) -> bytes:
        if self._bytes_repr is None:
            self._bytes_repr = self.values.tobytes()
        return self._bytes_repr

    @property
    def key(self) -> _RowKey:
        if self._key is None:
            self._key = _RowKey(self.values)
        return self._key

    def __eq__(self, other) -> bool:
        return self.bytes_repr == other.bytes_repr

    def update(self, a: int, o: int) -> 'Belief':
        key = f'{a}_{o}'
        hit = self._successors.get(key)
        if hit is not None:
            return hit
        m = self.model
        weights = m.reachable_transitional_observation_table[:, a, o, :] * self.values[:, None]
        nxt = np.bincount(m.reachable_states[:, a, :].flatten(), weights=weights.flatten(), minlength=m.state_count)
        nxt /= np.sum(nxt)
        out = self.__new__(self.__class__)
        out.model = m
        out._values = nxt
        self._successors[key] = out
        return out

    def generate_successors(self) -> list:
        return [self.update(a, o) for a in self.model.actions for o in self.model.observations]

    def random_state(self) -> int:
        return int(self.model.states[_draw_index(self._values)])


class BeliefSet:
    """A set of beliefs as a list and as a B x S matrix (``src/pomdp.py:489-659``)."""

    def __init__(self, model: Model, beliefs: Union[list, np.ndarray]) -> None:
        self.model = model
        self._belief_array = None
        self._uniqueness_dict = None
        self._key_dict = None
        self.is_on_gpu = bool(getattr(model, 'is_on_gpu', False))
        if isinstance(beliefs, list):
            assert all(len(b.values) == model.state_count for b in beliefs), \
                f"Beliefs in belief list provided dont all have shape ({model.state_count},)"
            self._belief_list = beliefs
        else:
            assert beliefs.shape[1] == model.state_count, \
                f"Belief array provided doesnt have the right shape (expected (-,{model.state_count}), received {beliefs.shape})"
            self._belief_list = [Belief(model, row) for row in beliefs]

    @property
    def belief_array(self) -> np.ndarray:
        if self._belief_array is None:
            self._belief_array = np.array([b.values for b in self._belief_list])
        return self._belief_array

    @property
    def belief_list(self) -> list:
        if self._belief_list is None:
            self._belief_list = [Belief(self.model, row) for row in self._belief_array]
        return self._belief_list

    def generate_all_successors(self) -> 'BeliefSet':
        succ = []
        for b in self.belief_list:
            succ.extend(b.generate_successors())
        return BeliefSet(self.model, succ)

    @property
    def unique_belief_dict(self) -> dict:
        """Beliefs keyed by their bytes, as the reference exposes them (built on demand)."""
        if self._uniqueness_dict is None:
            self._uniqueness_dict = {b.bytes_repr: b for b in self.belief_list}
        return self._uniqueness_dict

    @property
    def _unique_by_key(self) -> dict:
        if self._key_dict is None:
            self._key_dict = {b.key: b for b in self.belief_list}
        return self._key_dict

    def union(self, other: 'BeliefSet') -> 'BeliefSet':
        """``self.unique_belief_dict | other.unique_belief_dict`` (``src/pomdp.py:585-606``): this set's order, then
        the other's beliefs that are new by bytes; on equal bytes the other's object takes the slot.  Both inputs
        were validated when they were built, so the result is assembled directly; when every belief already lives
        in an engine's belief store (``_dev``), the id array of the result is carried over instead of re-walking
        the objects on the next device call."""
        mine = self._unique_by_key                           # same dedup as the bytes-keyed dictionaries, cheaper keys
        theirs = other._unique_by_key
        added = [b for k, b in theirs.items() if k not in mine]
        merged = dict(mine)
        merged.update(theirs)
        out = BeliefSet.__new__(BeliefSet)
        out.model = self.model
        out._belief_array = None
        out._uniqueness_dict = None
        out._key_dict = merged
        out.is_on_gpu = self.is_on_gpu
        out._belief_list = list(merged.values())
        ids = getattr(self, '_dev_ids', None)
        if ids is not None and len(ids[1]) == len(mine) and len(self._belief_list) == len(mine):
            # a slot the other set's object took over keeps its id: equal bytes, so it names an equal row
            tag = ids[0]
            if all(getattr(b, '_dev', (None,))[0] == tag for b in added):
                out._dev_ids = (tag, np.concatenate([ids[1], np.fromiter((b._dev[1] for b in added), dtype=np.int32,
                                                                          count=len(added))]))
        return out

    def __len__(self) -> int:
        return len(self._belief_list) if self._belief_list is not None else self._belief_array.shape[0]

    def to_gpu(self) -> 'BeliefSet':
        gm = self.model.gpu_model
        out = BeliefSet(gm, [Belief(gm, b.values) for b in self.belief_list])
        return out

    def to_cpu(self) -> 'BeliefSet':
        cm = self.model.cpu_model
        return BeliefSet(cm, [Belief(cm, b.values) for b in self.belief_list])


class BeliefValueMapping:
    """Upper bound of the value function as (belief, value) points over the corner values, evaluated with the
    sawtooth interpolation (``src/pomdp.py:786-895``; used by the HSVI expansion only).  Host-side: the points
    are a few hundred rows at most."""

    def __init__(self, model, corner_belief_values: ValueFunction) -> None:
        self.model = model
        self.corner_belief_values = corner_belief_values
        self.corner_values = np.max(corner_belief_values.alpha_vector_array, axis=0)
        self.beliefs = []
        self.belief_value_mapping = {}
        self._belief_array = None
        self._value_array = None

    def add(self, b: Belief, v: float) -> None:
        """Record ``v`` at ``b``; a belief that is already a point keeps its first value."""
        if b.bytes_repr not in self.belief_value_mapping:
            self.beliefs.append(b)
            self.belief_value_mapping[b.bytes_repr] = v

    def update(self) -> None:
        """Re-stack the cached point arrays (``evaluate`` reads the cache, so points added since the last call
        are not seen until this runs -- the reference's behaviour, ``src/pomdp.py:863-870``)."""
        self._belief_array = np.array([b.values for b in self.beliefs])
        self._value_array = np.array(list(self.belief_value_mapping.values()))

    @property
    def belief_array(self) -> np.ndarray:
        if self._belief_array is None:
            self._belief_array = np.array([b.values for b in self.beliefs])
        return self._belief_array

    @property
    def value_array(self) -> np.ndarray:
        if self._value_array is None:
            self._value_array = np.array(list(self.belief_value_mapping.values()))
        return self._value_array

    def evaluate(self, belief: Belief) -> float:
        hit = self.belief_value_mapping.get(belief.bytes_repr)
        if hit is not None:
            return hit
        v0 = np.dot(belief.values, self.corner_values)
        if len(self.beliefs) == 0:
            return float(v0)
        with np.errstate(divide='ignore', invalid='ignore'):
            gap = self.value_array - np.dot(self.belief_array, self.corner_values)
            vb = v0 + gap * np.min(belief.values / self.belief_array, axis=1)
        return float(np.min(np.append(vb, v0)))


class SolverHistory:
    """Times and sizes of a PBVI run (subset of ``src/pomdp.py:898-1117``)."""

    def __init__(self, tracking_level, model, gamma, eps, expand_function, expand_append,
                 initial_value_function, initial_belief_set):
        self.tracking_level = tracking_level
        self.model = model
        self.gamma = gamma
        self.eps = eps
        self.run_ts = datetime.now()
        self.expand_function = expand_function
        self.expand_append = expand_append
        self.expansion_times = []
        self.backup_times = []
        self.pruning_times = []
        self.alpha_vector_counts = []
        self.beliefs_counts = []
        self.value_function_changes = []
        self.prune_counts = []
        self.value_functions = []
        self.belief_sets = []
        if tracking_level >= 1:
            self.alpha_vector_counts.append(len(initial_value_function))
            self.beliefs_counts.append(len(initial_belief_set))
        if tracking_level >= 2:
            self.value_functions.append(initial_value_function)
            self.belief_sets.append(initial_belief_set)

    def add_expand_step(self, expansion_time: float, belief_set: BeliefSet) -> None:
        if self.tracking_level >= 1:
            self.expansion_times.append(float(expansion_time))
            self.beliefs_counts.append(len(belief_set))
        if self.tracking_level >= 2:
            self.belief_sets.append(belief_set)

    def add_backup_step(self, backup_time: float, value_function_change: float, value_function: ValueFunction) -> None:
        if self.tracking_level >= 1:
            self.backup_times.append(float(backup_time))
            self.alpha_vector_counts.append(len(value_function))
            self.value_function_changes.append(float(value_function_change))
        if self.tracking_level >= 2:
            self.value_functions.append(value_function)

    def add_prune_step(self, prune_time: float, alpha_vectors_pruned: int) -> None:
        if self.tracking_level >= 1:
            self.pruning_times.append(prune_time)
            self.prune_counts.append(alpha_vectors_pruned)

    @property
    def solution(self) -> ValueFunction:
        assert self.tracking_level >= 2, "Tracking level is set too low, increase it to 2 if you want to have value function tracking as well."
        return self.value_functions[-1]

    @property
    def explored_beliefs(self) -> BeliefSet:
        assert self.tracking_level >= 2, "Tracking level is set too low, increase it to 2 if you want to have belief sets tracking as well."
        return self.belief_sets[-1]

    @property
    def summary(self) -> str:
        n_b, n_e = len(self.backup_times), len(self.expansion_times)
        s = f'Summary of Point Based Value Iteration run\n'
        s += f'  - Model: {self.model.state_count} state, {self.model.action_count} action, {self.model.observation_count} observations\n'
        s += f'  - Converged or stopped after {n_e} expansion steps and {n_b} backup steps.\n'
        if n_b and n_e:
            s += f'  - Resulting value function has {self.alpha_vector_counts[-1]} alpha vectors.\n'
            s += f'  - Converged in {sum(self.expansion_times) + sum(self.backup_times):.4f}s\n\n'
            s += f'  - Expand function took on average {sum(self.expansion_times) / n_e:.4f}s\n'
            s += f'  - Backup function took on average {sum(self.backup_times) / n_b:.4f}s\n'
        return s


class Solver(MDP_Solver):
    def solve(self, model):
        raise Exception("Method has to be implemented by subclass...")


class PBVI_Solver(Solver):
    """Point-Based Value Iteration (``src/pomdp.py:1301-2413``).

    ``backup`` is the hot path.  On GPU-resident inputs it is one call into the
    HIP engine; on host inputs it is the reference's NumPy statement sequence.
    """

    def __init__(self, gamma: float = 0.99, eps: float = 0.001, expand_function: str = 'ssea', **expand_function_params):
        self.gamma = gamma
        self.eps = eps
        self.expand_function = expand_function
        self.expand_function_params = expand_function_params

    def test_n_simulations(self, model: Model, value_function: ValueFunction, n: int = 1000, horizon: int = 300,
                           print_progress: bool = False):
        """Roll the greedy policy of ``value_function`` out in ``n`` simulations advanced together
        (``src/pomdp.py:1338-1444``).  Unlike ``Agent.run_n_simulations_parallel`` every simulation keeps being
        stepped after it reached an end state (its rewards are masked instead).  Returns ``(start_states,
        done_at_step, rewards, discounted_rewards)`` -- the last two are lists of ``[n]`` arrays, one per step.
        With the value function on the GPU the belief block lives in the HIP engine for the whole run."""
        on_gpu = value_function.is_on_gpu
        model = model.gpu_model if on_gpu else model.cpu_model
        beliefs = np.repeat(Belief(model).values[None, :], n, axis=0)
        sims = SimulationSet(model.cpu_model)
        start_states = sims.initialize_simulations(n)
        block = (_DeviceBeliefBlock if on_gpu else _HostBeliefBlock)(model, value_function, beliefs)
        everyone = np.ones(n, dtype=bool)
        end_states = np.array(model.end_states)
        done_at_step = np.full(n, -1)
        actions_of = np.asarray(value_function.actions)
        discount = self.gamma
        rewards, discounted_rewards = [], []
        for i in range(horizon):
            was_done = sims.is_done.copy()
            best_actions = actions_of[block.best_vectors()]
            step_rewards, observations = sims.run_actions(best_actions)
            block.advance(best_actions, observations, everyone)
            rewards.append(step_rewards)
            discounted_rewards.append(step_rewards * discount)
            are_done = np.isin(sims.agent_states, end_states)
            done_at_step[was_done ^ are_done] = i + 1
            discount *= self.gamma
            if np.all(sims.is_done):
                break
        return start_states, done_at_step, rewards, discounted_rewards

    # ------------------------------------------------------------------ #
    # hot path
    # ------------------------------------------------------------------ #
    BELIEF_BLOCK = 32768      # beliefs per engine call on the GPU path
    shard_beliefs = None      # True: shard `backup` over the ranks of the torch.distributed job; None: dist.enable() / PBVI_SHARD

    def backup(self, model: Model, belief_set: BeliefSet, value_function: ValueFunction,
               append: bool = False, belief_dominance_prune: bool = True) -> ValueFunction:
        """One point-based backup (``src/pomdp.py:1447-1524``): B beliefs x V
        alpha-vectors -> at most B new alpha-vectors (+ union with the old set).

        With ``self.shard_beliefs = True`` (or ``dist.enable()`` / ``PBVI_SHARD=1``) as one rank of a multi-rank
        ``torch.distributed`` job (one process per GPU) the beliefs are sharded over the ranks, one all-gather makes
        every rank hold the whole result and every replica appends the same rows (``dist.sharded_backup``); the return
        value is the single-process one on every rank.  Opt-in: every rank must call with the same model, beliefs and
        value function (checked as far as a message trailer can: ``dist.ReplicaMismatch``)."""
        if _dist.active(solver=self) and len(belief_set) > 0:
            new_vf = _dist.sharded_backup(self, model, belief_set, value_function, belief_dominance_prune)
        elif value_function.is_on_gpu:
            # Residency: every AlphaVector / Belief row is uploaded once into the engine's device stores; the
            # working sets are selected by id in list order (tie-breaks follow the host order), the device
            # runs the backup, and only the distinct new rows come back.
            eng = value_function.model.engine
            beliefs = belief_set.belief_list
            try:
                if self._belief_chunk is not None and len(beliefs) > self._belief_chunk:
                    raise MemoryError('an earlier backup of this solver only fitted in belief chunks')
                alpha_new, actions = self._backup_block(eng, value_function, belief_set, beliefs, belief_dominance_prune)
            except MemoryError as e:
                if len(beliefs) < 2:
                    raise
                if self._belief_chunk is None:
                    log(f'[Warning] Backup of {len(beliefs)} beliefs does not fit the device ({e}); continuing in belief chunks...')
                alpha_new, actions = self._backup_in_chunks(eng, value_function, beliefs, belief_dominance_prune)
            new_vf = (ValueFunction(value_function.model, alpha_new) if isinstance(alpha_new, list)
                      else ValueFunction(value_function.model, alpha_new, actions))
        else:
            alpha_new, actions = self._backup_numpy(model, belief_set.belief_array, value_function.alpha_vector_array,
                                                    belief_dominance_prune)
            new_vf = ValueFunction(model, alpha_new, actions)
        if append:
            new_vf.extend(value_function)
        return new_vf

    _belief_chunk = None      # set once a block only fitted the device in pieces: later backups start there

    def _backup_block(self, eng, value_function, belief_set, beliefs, belief_dominance_prune):
        """The device backup of one belief list against the resident value function: rows (``AlphaVector`` objects tagged
        with their device residency, or an array) and actions."""
        eng.sync_rows('alpha', value_function.alpha_vector_list, lambda v: v.values, owner=value_function)
        if len(beliefs) <= self.BELIEF_BLOCK:
            eng.sync_rows('belief', beliefs, lambda b: b.values, owner=belief_set)
            stats = eng.run(self.gamma, belief_dominance_prune)
            if stats['formulation'] == 2:
                # belief-side GEMM: it also multiplied the beliefs themselves by the alpha set; compute_change asks for
                # exactly those maxima next (these beliefs, the value function that was just backed up)
                eng.seed_max_values(value_function.alpha_vector_list, beliefs, lambda v: v.values, lambda b: b.values,
                                    alpha_owner=value_function, belief_owner=belief_set)
            alpha_new, actions, uidx = eng.fetch().value_function_rows(use_keep=belief_dominance_prune, with_index=True)
            if len(uidx):
                # the new rows join the engine's alpha store device to device: the next call selects them by id;
                # their dedup hashes come from the device too (ValueFunction keys its dictionary on them)
                first, tag = eng.store_unique(uidx), eng.store_tag('alpha')
                hashes = eng.fetch_row_hashes()[uidx]
                vectors = []
                for k, (row, act) in enumerate(zip(alpha_new, actions)):
                    v = AlphaVector(row, act)
                    v._dev = (tag, first + k)
                    v._hash = int(hashes[k])
                    vectors.append(v)
                alpha_new = vectors
        else:   # beliefs are independent: larger sets go through the engine in blocks (it takes 65535 at a time)
            parts = []
            for i0 in range(0, len(beliefs), self.BELIEF_BLOCK):
                eng.sync_rows('belief', beliefs[i0:i0 + self.BELIEF_BLOCK], lambda b: b.values)
                eng.run(self.gamma, belief_dominance_prune)
                parts.append(eng.fetch().value_function_rows(use_keep=belief_dominance_prune))
            alpha_new = np.concatenate([p[0] for p in parts])
            actions = np.concatenate([p[1] for p in parts])
        return alpha_new, actions

    def _backup_in_chunks(self, eng, value_function, beliefs, belief_dominance_prune):
        """A backup whose working set did not fit the device (the engine raised ``MemoryError`` and is back in its freshly
        created state), done in belief chunks: beliefs are independent, so the union of the chunks' rows is the block's
        result (the constructor of ``ValueFunction`` drops duplicates as it does for one block).  Chunks project the
        BELIEFS through the model, whose footprint is chunk x A x O rows, instead of Gamma's A x O x V rows -- the array
        the reference's CuPy path dies allocating (``Sea_Robin_Real.ipynb:913``) -- and are halved until one fits; a
        single belief that does not fit re-raises, which ``solve`` turns into the partial result
        (``src/pomdp.py:2399-2401``)."""
        chunk = self._belief_chunk if self._belief_chunk is not None else (len(beliefs) + 1) // 2
        chunk = max(1, min(chunk, (len(beliefs) + 1) // 2))
        setting = eng.formulation
        while True:
            # (not inside the try: a value function that does not fit by itself is not cured by smaller chunks)
            eng.sync_rows('alpha', value_function.alpha_vector_list, lambda v: v.values, owner=value_function)
            try:
                eng.set_formulation('belief' if chunk <= len(value_function) else 'auto')
                parts = []
                for i0 in range(0, len(beliefs), chunk):
                    eng.sync_rows('belief', beliefs[i0:i0 + chunk], lambda b: b.values)
                    eng.run(self.gamma, belief_dominance_prune)
                    parts.append(eng.fetch().value_function_rows(use_keep=belief_dominance_prune))
                self._belief_chunk = chunk
                return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
            except MemoryError:
                if chunk == 1:
                    raise
                chunk = (chunk + 1) // 2
            finally:
                eng.set_formulation(setting)

    def _backup_numpy(self, model, b, alpha, belief_dominance_prune, return_mask: bool = False):
        """Host path: the reference's array statements (``src/pomdp.py:1485-1515``).  ``return_mask``: every belief's
        row and action plus the belief-dominance mask, instead of the filtered rows (the sharded backup exchanges
        per-belief results)."""
        V = alpha.shape[0]
        alpha_r = alpha[np.arange(V)[:, None, None, None], model.reachable_states[None, :, :, :]]
        gamma_aovs = self.gamma * np.einsum('saor,vsar->aovs', model.reachable_transitional_observation_table, alpha_r)
        pick = np.argmax(np.tensordot(b, gamma_aovs, (1, 3)), axis=3)
        per_o = gamma_aovs[model.actions[None, :, None, None], model.observations[None, None, :, None],
                           pick[:, :, :, None], model.states[None, None, None, :]]
        alpha_a = model.expected_rewards_table.T + np.sum(per_o, axis=2)
        acts = np.argmax(np.einsum('bas,bs->ba', alpha_a, b), axis=1)
        rows = np.take_along_axis(alpha_a, acts[:, None, None], axis=1)[:, 0, :]
        better = np.ones(len(acts), dtype=bool)
        if belief_dominance_prune:
            new_val = np.sum(b * rows, axis=1)
            old_val = np.max(np.matmul(b, alpha.T), axis=1)
            better = new_val > old_val
        if return_mask:
            return rows, acts, better
        if belief_dominance_prune:
            rows, acts = rows[better], acts[better]
        return rows, acts

    # ------------------------------------------------------------------ #
    # belief expansion (host side; out of the accelerated path)
    # ------------------------------------------------------------------ #
    def expand_ra(self, model, belief_set, max_generation: int = 10) -> BeliefSet:
        n = min(belief_set.belief_array.shape[0], max_generation)
        nb = np.random.random((n, model.state_count))
        nb /= np.sum(nb, axis=1)[:, None]
        return BeliefSet(model, nb)

    def _sim_step(self, model, b: Belief, a: int) -> Belief:
        s = b.random_state()
        s_p = model.transition(s, a)
        return b.update(a, model.observe(s_p, a))

    def expand_ssra(self, model, belief_set, max_generation: int = 10) -> BeliefSet:
        arr = belief_set.belief_array
        n = min(max_generation, arr.shape[0])
        picks = np.random.choice(np.arange(arr.shape[0]), n, replace=False)
        out = np.empty((n, arr.shape[1]))
        for i, row in enumerate(arr[picks]):
            b = Belief(model, row)
            s = b.random_state()
            a = random.choice(model.actions)
            s_p = model.transition(s, a)
            out[i] = b.update(a, model.observe(s_p, a)).values
        return BeliefSet(model, out)

    def expand_ssga(self, model, belief_set, value_function, epsilon: float = 0.1, max_generation: int = 10) -> BeliefSet:
        arr = belief_set.belief_array
        n = min(max_generation, arr.shape[0])
        picks = np.random.choice(np.arange(arr.shape[0]), n, replace=False)
        out = np.empty((n, arr.shape[1]))
        for i, row in enumerate(arr[picks]):
            b = Belief(model, row)
            s = b.random_state()
            if random.random() < epsilon:
                a = random.choice(model.actions)
            else:
                a = value_function.actions[np.argmax(np.dot(value_function.alpha_vector_array, b.values))]
            s_p = model.transition(s, a)
            out[i] = b.update(a, model.observe(s_p, a)).values
        return BeliefSet(model, out)

    def expand_ssea(self, model, belief_set, max_generation: int = 10) -> BeliefSet:
        arr = belief_set.belief_array
        n = min(max_generation, arr.shape[0])
        succ = np.array([[[b.update(a, o).values for o in model.observations] for a in model.actions]
                         for b in belief_set.belief_list])
        diff = arr[:, None, None, None, :] - succ
        dist = np.sqrt(np.einsum('bnaos,bnaos->bnao', diff, diff))
        nearest = np.min(dist, axis=0)
        b_i, a_i, o_i = np.unravel_index(np.argsort(nearest, axis=None)[::-1][:n], succ.shape[:-1])
        return BeliefSet(model, succ[b_i[:, None], a_i[:, None], o_i[:, None], model.states[None, :]])

    def expand_ger(self, model, belief_set, value_function, max_generation: int = 10) -> BeliefSet:
        arr = belief_set.belief_array
        n = min(max_generation, arr.shape[0])
        r_lo = model._min_reward / (1 - self.gamma)
        r_hi = model._max_reward / (1 - self.gamma)
        succ = np.array([[[b.update(a, o).values for o in model.observations] for a in model.actions]
                         for b in belief_set.belief_list])
        va = value_function.alpha_vector_array
        own = va[np.argmax(np.dot(arr, va.T), axis=1)]
        d_b = succ - arr[:, None, None, :]
        d_a = np.where(d_b >= 0, r_hi, r_lo) - own[:, None, None, :]
        err = np.einsum('baos,baos->bao', d_a, d_b)
        p_bao = np.einsum('bs,saor->bao', arr, model.reachable_transitional_observation_table)
        score = np.einsum('bao,bao->ba', p_bao, err)
        b_i, a_i = np.unravel_index(np.argsort(score, axis=None)[::-1][:n], score.shape)
        o_i = np.argmax(p_bao[b_i[:, None], a_i[:, None], model.observations[None, :]]
                        * err[b_i[:, None], a_i[:, None], model.observations[None, :]], axis=1)
        return BeliefSet(model, succ[b_i[:, None], a_i[:, None], o_i[:, None], model.states[None, :]])

    def expand_hsvi(self, model, b: Belief, value_function: ValueFunction, upper_bound_belief_value_map: BeliefValueMapping,
                    conv_term: Union[float, None] = None, max_generation: int = 10) -> BeliefSet:
        """HSVI descent (``src/pomdp.py:1768-1868``): from ``b`` take the action that is greedy for the upper bound,
        then the observation with the largest probability-weighted gap between the bounds, until the gap falls under
        ``eps / gamma^depth`` or ``max_generation`` beliefs are out.  The deepest belief comes first in the result.
        The lower bound ``max_v alpha_v . b`` is one small matrix-vector product per observation and stays on the host."""
        rto = model.reachable_transitional_observation_table
        alpha = value_function.alpha_vector_array
        conv = self.eps if conv_term is None else conv_term
        chain = []
        while True:
            conv /= self.gamma
            best_q, best_a = -np.inf, -1
            for a in model.actions:
                p_o = np.einsum('sor,s->o', rto[:, a, :, :], b.values)
                tail = 0
                for o in model.observations:
                    tail += p_o[o] * upper_bound_belief_value_map.evaluate(b.update(a, o))
                q = float(np.dot(model.expected_rewards_table[:, a], b.values) + self.gamma * tail)
                if q > best_q:
                    best_q, best_a = q, a
            p_o = np.einsum('sor,s->o', rto[:, best_a, :, :], b.values)
            top, gap_at_top, nxt = -np.inf, -np.inf, b
            for o in model.observations:
                bao = b.update(best_a, o)
                gap = upper_bound_belief_value_map.evaluate(bao) - np.max(np.dot(alpha, bao.values))
                if p_o[o] * gap > top:
                    top, gap_at_top, nxt = p_o[o] * gap, gap, bao
            chain.append(nxt)
            if gap_at_top < conv or max_generation <= 1:
                break
            upper_bound_belief_value_map.add(b, best_q)
            b = nxt
            max_generation -= 1
        return BeliefSet(model, chain[::-1])

    def _walk_device(self, model, b0: Belief, policy_action, max_generation: int) -> BeliefSet:
        """``_walk`` with the beliefs on the device.  The (action, observation) trajectory is simulated in the underlying
        MDP and does not depend on the beliefs, so it is drawn first (same random draws, same order), then the n
        chained Bayes updates run in one C-ABI call (``pbvi_belief_walk``): the new beliefs land in the engine's
        belief store -- the following backup selects them by id, nothing is uploaded -- and their fp64 values come
        back once for the containers' byte-keyed dedup."""
        acts, obs, restart = [], [], []
        s = b0.random_state()
        fresh = True                                   # the belief this step starts from is b0
        for i in range(max_generation - 1):
            a = int(policy_action(i, s))
            s_p = model.transition(s, a)
            acts.append(a)
            obs.append(model.observe(s_p, a))
            restart.append(fresh)
            fresh = False
            s = s_p
            if s in model.end_states:
                s = b0.random_state()
                fresh = True
        if not acts:
            return BeliefSet(model, [b0])
        eng = model.engine
        values, first = eng.belief_walk(b0.values, acts, obs, restart)
        sums = eng.belief_walk_keys(len(acts)).tolist() if hasattr(eng, 'belief_walk_keys') else None
        tag = eng.belief_tag()
        seq = [b0]
        for i in range(len(acts)):
            nb = Belief.__new__(Belief)
            nb.model = b0.model
            nb._values = values[i]
            nb._dev = (tag, first + i)
            if sums is not None:                       # the dedup key's hash comes from the device: no host pass over the row
                nb._key = _RowKey.from_sum(sums[i], values[i])
            seq.append(nb)
        return BeliefSet(model, seq)

    def _walk(self, model, b0: Belief, policy_action, max_generation: int) -> BeliefSet:
        if getattr(model, 'is_on_gpu', False):
            return self._walk_device(model, b0, policy_action, max_generation)
        seq = [b0]
        s = b0.random_state()
        b = b0
        for i in range(max_generation - 1):
            a = policy_action(i, s)
            s_p = model.transition(s, a)
            b = b.update(a, model.observe(s_p, a))
            seq.append(b)
            s = s_p
            if s in model.end_states:
                s = b0.random_state()
                b = b0
        return BeliefSet(model, seq)

    def expand_fsvi(self, model, b0: Belief, mdp_policy: ValueFunction, max_generation: int = 10) -> BeliefSet:
        q = mdp_policy.alpha_vector_array
        return self._walk(model, b0, lambda i, s: np.argmax(q[:, s]), max_generation)

    def expand_fsvi_eg(self, model, b0, mdp_policy, eps_greedy=None, max_generation: int = 10) -> BeliefSet:
        q = mdp_policy.alpha_vector_array
        eg = eps_greedy if eps_greedy is not None else (lambda t: 0.2)
        return self._walk(model, b0, lambda i, s: int(random.choice(model.actions)) if random.random() < eg(i)
                          else np.argmax(q[:, s]), max_generation)

    def expand_perseus(self, model, b: Belief, max_generation: int = 10) -> BeliefSet:
        seq = []
        for _ in range(max_generation):
            a = int(np.random.choice(model.actions, size=1)[0])
            p_o = np.einsum('sor,s->o', model.reachable_transitional_observation_table[:, a, :, :], b.values)
            o = int(np.random.choice(model.observations, size=1, p=p_o)[0])
            b = b.update(a, o)
            seq.append(b)
        return BeliefSet(model, seq)

    def expand(self, model, belief_set, max_generation: int, **params) -> BeliefSet:
        """Dispatch by substring exactly like ``src/pomdp.py:2088-2136``."""
        f = self.expand_function
        if f in 'expand_ra':
            return self.expand_ra(model, belief_set, max_generation)
        if f in 'expand_ssra':
            return self.expand_ssra(model, belief_set, max_generation)
        if f in 'expand_ssga':
            kw = {k: params[k] for k in ('value_function', 'epsilon') if k in params}
            return self.expand_ssga(model, belief_set, max_generation=max_generation, **kw)
        if f in 'expand_ssea':
            return self.expand_ssea(model, belief_set, max_generation)
        if f in 'expand_ger':
            return self.expand_ger(model, belief_set, params['value_function'], max_generation)
        if f in 'expand_hsvi':
            if not hasattr(self, '_upper_bound'):          # kept on the solver across calls, like src/pomdp.py:2112-2115
                self._upper_bound = BeliefValueMapping(model, params['mdp_policy'])
            else:
                self._upper_bound.update()
            return self.expand_hsvi(model, belief_set.belief_list[0], params['value_function'], self._upper_bound,
                                    max_generation=max_generation)
        if f in 'expand_fsvi':
            return self.expand_fsvi(model, belief_set.belief_list[0], params['mdp_policy'], max_generation)
        if f in 'expand_fsvi_eg':
            return self.expand_fsvi_eg(model, belief_set.belief_list[0], params['mdp_policy'],
                                       params.get('eps_greedy'), max_generation)
        if f in 'expand_perseus':
            return self.expand_perseus(model, belief_set.belief_list[0], max_generation)
        raise Exception('Not implemented')

    def compute_change(self, value_function: ValueFunction, new_value_function: ValueFunction, belief_set: BeliefSet) -> float:
        """Largest change of ``max_v b.alpha_v`` over the belief set (``src/pomdp.py:2141-2169``)."""
        if value_function.is_on_gpu:
            # the belief set and the alpha set only grow between calls: the engine wrapper scores just the new
            # (belief, alpha) pairs and keeps the rest (Engine.max_value_objects)
            eng = value_function.model.engine
            beliefs = belief_set.belief_list
            # smaller alpha set first: in the solve loop it is the previous value function, a subset of the other one,
            # so the larger set then only needs (all beliefs x its additional rows)
            pair = sorted((value_function, new_value_function), key=len)
            # only max|new - old| of the values is used, against eps * gamma / (1 - gamma): an f32 engine returns its
            # GEMM's maxima as they are (exact=False); f64 engines are exact either way
            vals = {id(vf): eng.max_value_objects(vf.alpha_vector_list, beliefs, lambda v: v.values, lambda x: x.values,
                                                  alpha_owner=vf, belief_owner=belief_set, exact=False) for vf in pair}
            old, new = vals[id(value_function)], vals[id(new_value_function)]
        else:
            b = belief_set.belief_array
            old = np.max(np.matmul(b, value_function.alpha_vector_array.T), axis=1)
            new = np.max(np.matmul(b, new_value_function.alpha_vector_array.T), axis=1)
        return float(np.max(np.abs(new - old)))

    def _limit_value_function(self, model, value_function: ValueFunction, belief_set: BeliefSet,
                              max_belief_growth: int) -> ValueFunction:
        """The |V| limiter of the solve loop (``src/pomdp.py:2347-2365``): alpha-vectors that are the best one for no
        belief of the set are "unuseful"; ``max_belief_growth`` of them, drawn with replacement and weights falling
        linearly with their position, are deleted.  The usefulness scan -- ``argmax_v`` of the V x B_total score matrix --
        runs on the engine when the value function is on the GPU (``pbvi_value_max_store`` over the belief store in
        place, first maximum, exact indices); the draw stays ``np.random.choice`` on the host, as in the reference."""
        n = len(value_function)
        if value_function.is_on_gpu:
            eng = value_function.model.engine
            beliefs = belief_set.belief_list
            eng.sync_rows('alpha', value_function.alpha_vector_list, lambda v: v.values, owner=value_function)
            b_ids = eng.row_ids('belief', beliefs, lambda b: b.values, owner=belief_set)
            best = eng.best_alpha_of_store_rows(b_ids)
        else:
            best = np.argmax(np.matmul(value_function.alpha_vector_array, belief_set.belief_array.T), axis=0)
        useful = np.unique(best)
        useless = np.delete(np.arange(n), useful)
        w = np.arange(len(useless))[::-1]
        drop = np.random.choice(useless, size=max_belief_growth, p=w / np.sum(w))
        self._last_useful_count = int(useful.shape[0])
        if value_function.is_on_gpu:
            # on the objects: the surviving vectors keep their rows in the device store, nothing is re-stacked
            gone = set(int(i) for i in drop)
            return ValueFunction(value_function.model, [v for i, v in enumerate(value_function.alpha_vector_list) if i not in gone])
        return ValueFunction(model, np.delete(value_function.alpha_vector_array, drop, axis=0),
                             np.delete(value_function.actions, drop))

    def solve(self, model: Model, expansions: int, full_backup: Union[bool, None] = None, update_passes: int = 1,
              max_belief_growth: int = 10, initial_belief=None, initial_value_function=None, prune_level: int = 1,
              prune_interval: int = 10, limit_value_function_size: int = -1, use_gpu: bool = False,
              history_tracking_level: int = 1, print_progress: bool = True, engine_dtype: str = 'f64'):
        """Expand / backup loop (``src/pomdp.py:2172-2413``).  ``engine_dtype``
        ('f64' or 'f32') is the one added keyword: the arithmetic type of the HIP
        engine when ``use_gpu=True``."""
        if use_gpu:
            model = model.to_gpu(engine_dtype) if not model.is_on_gpu else model

        if initial_belief is None:
            belief_set = BeliefSet(model, [Belief(model)])
        elif isinstance(initial_belief, BeliefSet):
            belief_set = initial_belief.to_gpu() if use_gpu else initial_belief
        else:
            belief_set = BeliefSet(model, [Belief(model, np.array(initial_belief.values))])

        if initial_value_function is None:
            value_function = ValueFunction(model, model.expected_rewards_table.T, model.actions)
        else:
            value_function = initial_value_function.to_gpu() if use_gpu else initial_value_function

        if full_backup is None:
            full_backup = any(self.expand_function in f for f in
                              ['expand_ra', 'expand_ssra', 'expand_ssga', 'expand_ssea', 'expand_ger'])

        if ('fsvi' in self.expand_function or 'hsvi' in self.expand_function) and \
                self.expand_function_params.get('mdp_policy') is None:
            log('[Warning] MDP solution not provided, running value iteration on the problem to retrieve it...')
            # Device sweeps are bit-identical to the NumPy loop for R = 1 only; with several reachable states the sum
            # over r may associate differently, and FSVI breaks exact Q-value ties (symmetric grids) by argmax --
            # keep the reference's arithmetic there so seeded runs follow the reference's trajectory.
            mdp_solution, _ = VI_Solver(gamma=self.gamma, eps=self.eps).solve(
                model, use_gpu=use_gpu and model.reachable_state_count == 1, print_progress=False)
            self.expand_function_params['mdp_policy'] = mdp_solution

        max_allowed_change = self.eps * (self.gamma / (1 - self.gamma))
        history = SolverHistory(history_tracking_level, model, self.gamma, self.eps, self.expand_function,
                                full_backup, value_function, belief_set)

        iteration = 0
        expand_value_function = value_function
        old_value_function = value_function
        self._belief_chunk = None              # (backup: the chunk size an earlier solve settled on says nothing about this one)
        try:
            for expansion_i in range(expansions):
                t0 = datetime.now()
                new_belief_set = self.expand(model=model, belief_set=belief_set, value_function=value_function,
                                             max_generation=max_belief_growth, **self.expand_function_params)
                belief_set = belief_set.union(new_belief_set)
                history.add_expand_step((datetime.now() - t0).total_seconds(), belief_set)

                for _ in range(update_passes):
                    t0 = datetime.now()
                    value_function = self.backup(model, belief_set if full_backup else new_belief_set, value_function,
                                                 append=(not full_backup), belief_dominance_prune=False)
                    backup_time = (datetime.now() - t0).total_seconds()

                    if (iteration % prune_interval) == 0 and iteration > 0:
                        t0 = datetime.now()
                        before = len(value_function)
                        value_function.prune(prune_level)
                        history.add_prune_step((datetime.now() - t0).total_seconds(), len(value_function) - before)

                    if limit_value_function_size >= 0 and len(value_function) > limit_value_function_size:
                        value_function = self._limit_value_function(model, value_function, belief_set, max_belief_growth)

                    max_change = self.compute_change(value_function, old_value_function, belief_set)
                    history.add_backup_step(backup_time, max_change, value_function)
                    if max_change < max_allowed_change:
                        break
                    old_value_function = value_function
                    iteration += 1

                if self.compute_change(expand_value_function, value_function, belief_set) < max_allowed_change:
                    print('Converged!')
                    break
                expand_value_function = value_function
        except MemoryError as e:
            print(f'Memory full: {e}')
            print('Returning value function and history as is...\n')

        t0 = datetime.now()
        before = len(value_function)
        value_function.prune(prune_level)
        history.add_prune_step((datetime.now() - t0).total_seconds(), len(value_function) - before)
        return value_function, history


class FSVI_Solver(PBVI_Solver):
    """Forward Search Value Iteration preset (``src/pomdp.py:2470-2544``)."""

    def __init__(self, gamma: float = 0.99, eps: float = 0.001, mdp_policy: Union[ValueFunction, None] = None):
        super().__init__(gamma, eps, 'fsvi', mdp_policy=mdp_policy)

    def solve(self, model, expansions, update_passes: int = 1, max_belief_growth: int = 10, initial_belief=None,
              initial_value_function=None, prune_level: int = 1, prune_interval: int = 10,
              limit_value_function_size: int = -1, use_gpu: bool = False, history_tracking_level: int = 1,
              print_progress: bool = True, engine_dtype: str = 'f64'):
        return super().solve(model=model, expansions=expansions, full_backup=False, update_passes=update_passes,
                             max_belief_growth=max_belief_growth, initial_belief=initial_belief,
                             initial_value_function=initial_value_function, prune_level=prune_level,
                             prune_interval=prune_interval, limit_value_function_size=limit_value_function_size,
                             use_gpu=use_gpu, history_tracking_level=history_tracking_level,
                             print_progress=print_progress, engine_dtype=engine_dtype)


class FSVI_EG_Solver(FSVI_Solver):
    """Epsilon-greedy FSVI preset (``src/pomdp.py:2547-2578``)."""

    def __init__(self, gamma: float = 0.99, eps: float = 0.001, mdp_policy=None, eps_greedy=None):
        PBVI_Solver.__init__(self, gamma, eps, 'fsvi_eg', mdp_policy=mdp_policy, eps_greedy=eps_greedy)


class HSVI_Solver(PBVI_Solver):
    """Heuristic Search Value Iteration preset (``src/pomdp.py:2416-2478``)."""

    def __init__(self, gamma: float = 0.99, eps: float = 0.001, mdp_solution: Union[ValueFunction, None] = None):
        super().__init__(gamma, eps, 'hsvi', mdp_policy=mdp_solution)

    def solve(self, model, expansions, max_belief_growth: int = 10, initial_belief=None, initial_value_function=None,
              prune_level: int = 1, prune_interval: int = 10, limit_value_function_size: int = -1, use_gpu: bool = False,
              history_tracking_level: int = 1, print_progress: bool = True, engine_dtype: str = 'f64'):
        return super().solve(model=model, expansions=expansions, full_backup=False, update_passes=1,
                             max_belief_growth=max_belief_growth, initial_belief=initial_belief,
                             initial_value_function=initial_value_function, prune_level=prune_level,
                             prune_interval=prune_interval, limit_value_function_size=limit_value_function_size,
                             use_gpu=use_gpu, history_tracking_level=history_tracking_level,
                             print_progress=print_progress, engine_dtype=engine_dtype)


# --------------------------------------------------------------------------- #
# Policy evaluation: simulators and the agent
# --------------------------------------------------------------------------- #
class SimulationHistory(MDP_SimulationHistory):
    """Episode record with observations and beliefs (``src/pomdp.py:2581-2662``).  The belief sequence is rebuilt
    on demand from the (action, observation) pairs when it was not recorded step by step."""

    def __init__(self, model: Model, start_state: int, start_belief: Belief):
        super().__init__(model, start_state)
        self._beliefs = [start_belief]
        self.observations = []

    @property
    def beliefs(self) -> list:
        if len(self._beliefs) < len(self):
            b = self._beliefs[0]
            self._beliefs = [b]
            for a, o in zip(self.actions, self.observations):
                b = b.update(int(a), int(o))
                self._beliefs.append(b)
        return self._beliefs

    def add(self, action: int, reward, next_state: int, next_belief: Belief, observation: int) -> None:
        super().add(action, reward, next_state)
        self._beliefs.append(next_belief)
        self.observations.append(observation)

    def to_dataframe(self, include_beliefs: bool = False):
        """The MDP columns plus ``Observations`` and, on request, one ``B_<state>`` column per state
        (``src/pomdp.py:2664-2686``)."""
        import pandas as pd
        df = super().to_dataframe()
        df['Observations'] = list(self.observations) + [None]
        if include_beliefs:
            rows = np.array([b.values.tolist() for b in self.beliefs])
            df = pd.concat([df, pd.DataFrame(rows, columns=[f'B_{sl}' for sl in self.model.state_labels])], axis=1)
        return df

    def save(self, path: str = './Simulations', file_name: Union[str, None] = None, include_beliefs: bool = False) -> None:
        target = self._csv_target(path, file_name)
        if not include_beliefs:
            print('[Warning] Beliefs not saved with simulation history but the belief sequence can be recreated from '
                  'the actions and observations.')
        self.to_dataframe(include_beliefs=include_beliefs).to_csv(target, index=False)
        print(f'Saved to: {target}')


class Simulation(MDP_Simulation):
    """One hidden-state walk with observations (``src/pomdp.py:2756-2815``)."""

    def __init__(self, model: Model) -> None:
        super().__init__(model)
        self.model = model

    def run_action(self, a: int) -> Tuple[Union[int, float], int]:
        assert not self.is_done, "Action run when simulation is done."
        s = self.agent_state
        s_p = self.model.transition(s, a)
        o = self.model.observe(s_p, a)
        r = self.model.reward(s, a, s_p, o)
        self.agent_state = s_p
        self._mark_done(s_p, a)
        return r, o


class SimulationSet:
    """n hidden-state walks advanced together (``src/pomdp.py:2818-2945``).  The random draws are NumPy's global
    stream in the reference's call order (start states: one ``choice``; per step: one ``choice`` per
    simulation when R > 1, then one ``random(n)``), so a seeded run reproduces the reference's trajectories.

    ``reference_indexing``: for R > 1 the reference picks the successor with ``potentials[chosen][:, 0, 0]``
    (``:2928``), i.e. row ``chosen[i]`` column 0 rather than row ``i`` column ``chosen[i]``.  The default keeps
    that behaviour for parity; set it to False to sample ``rs[s_i, a_i, chosen_i]``.
    """

    reference_indexing = True

    def __init__(self, model: Model):
        self.model = model
        self.n = -1
        self.agent_states = [-1]
        self.simulations = []
        self.is_done = [True]

    def initialize_simulations(self, n: int = 1, start_state: Union[list, int, None] = None) -> np.ndarray:
        if isinstance(start_state, int):
            states = (np.ones(n) * start_state).astype(int)
        elif isinstance(start_state, list):
            rep = np.repeat(np.array(start_state), int(np.ceil(n / len(start_state))))
            states = np.resize(rep, n)
        else:
            states = np.random.choice(self.model.states, size=n, p=self.model.start_probabilities).astype(int)
        self.n = n
        self.agent_states = states
        self.simulations = np.arange(n)
        self.is_done = np.zeros(n, dtype=bool)
        return self.agent_states

    def _step_rewards(self, s, a, s_p, o) -> np.ndarray:
        m = self.model
        if m.immediate_reward_table is not None:
            return np.asarray(m.immediate_reward_table[s, a, s_p, o])
        fn = m.immediate_reward_function
        if getattr(fn, '__func__', None) is Model._end_reward_function:      # element-wise by construction
            return np.asarray(fn(s, a, s_p, o))
        return np.array([fn(int(w), int(x), int(y), int(z)) for w, x, y, z in zip(s, a, s_p, o)])

    def run_actions(self, actions: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        m = self.model
        actions = np.asarray(actions)
        potentials = m.reachable_states[self.agent_states, actions]                 # [n, R]
        if m.reachable_state_count == 1:
            next_states = potentials[:, 0]
        else:
            probs = m.reachable_probabilities[self.agent_states, actions]            # [n, R]
            chosen = np.apply_along_axis(lambda x: np.random.choice(len(x), size=1, p=x), axis=1, arr=probs)
            if self.reference_indexing:
                next_states = potentials[chosen][:, 0, 0]
            else:
                next_states = potentials[np.arange(self.n), chosen[:, 0]]
        obs_p = m.observation_table[next_states, actions]                            # [n, O]
        observations = np.sum(np.random.random(self.n)[:, None] > np.cumsum(obs_p[:, :-1], axis=1), axis=1)
        step_rewards = self._step_rewards(self.agent_states, actions, next_states, observations)
        rewards = np.where(~self.is_done, step_rewards, 0)
        self.is_done |= np.isin(next_states, np.array(m.end_states))
        self.agent_states = next_states
        return rewards, observations


class _HostBeliefBlock:
    """Belief block of the parallel simulator held in NumPy (reference CPU statements, ``:3029``, ``:3306-3311``)."""

    def __init__(self, model: Model, value_function: ValueFunction, beliefs: np.ndarray):
        self.m, self.alpha, self.b = model, value_function.alpha_vector_array, beliefs

    def best_vectors(self) -> np.ndarray:
        return np.argmax(np.matmul(self.b, self.alpha.T), axis=1)

    def advance(self, actions: np.ndarray, observations: np.ndarray, keep: np.ndarray) -> None:
        m, n = self.m, self.b.shape[0]
        S = m.state_count
        w = m.reachable_transitional_observation_table[:, actions, observations, :] * self.b.T[:, :, None]   # [S,n,R]
        tgt = m.reachable_states[:, actions, :]                                                             # [S,n,R]
        flat = (n, S * m.reachable_state_count)
        idx = tgt.swapaxes(0, 1).reshape(flat) + (np.arange(n)[:, None] * S)
        nb = np.bincount(idx.ravel(), weights=w.swapaxes(0, 1).reshape(flat).ravel(), minlength=n * S).reshape((-1, S))
        nb /= np.sum(nb, axis=1)[:, None]
        self.b = nb[keep]


class _DeviceBeliefBlock:
    """Belief block resident in the HIP engine: ``pbvi_value_max`` + ``pbvi_beliefs_advance``.

    The engine holds one block of at most 65535 beliefs.  Larger simulations (the reference takes any ``n``) run in
    chunks of ``CHUNK`` rows that live on the host between steps and pass through the engine one after the other --
    correct, but every step then moves the beliefs over PCIe; the resident path is the one that is fast."""

    CHUNK = 32768

    def __init__(self, model: Model, value_function: ValueFunction, beliefs: np.ndarray):
        self.eng = model.engine
        self.eng.sync_rows('alpha', value_function.alpha_vector_list, lambda v: v.values)
        self.chunks = None
        if beliefs.shape[0] <= 65535:
            self.eng.set_beliefs(beliefs)
        else:
            self.chunks = [np.ascontiguousarray(beliefs[i:i + self.CHUNK]) for i in range(0, beliefs.shape[0], self.CHUNK)]

    def best_vectors(self) -> np.ndarray:
        if self.chunks is None:
            return self.eng.max_value_resident()[1]
        out = []
        for c in self.chunks:
            self.eng.set_beliefs(c)
            out.append(self.eng.max_value_resident()[1])
        return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)

    def advance(self, actions: np.ndarray, observations: np.ndarray, keep: np.ndarray) -> None:
        if self.chunks is None:
            self.eng.advance_beliefs(actions, observations, keep)
            return
        nxt, i0 = [], 0
        for c in self.chunks:
            n = c.shape[0]
            self.eng.set_beliefs(c)
            if self.eng.advance_beliefs(actions[i0:i0 + n], observations[i0:i0 + n], keep[i0:i0 + n]) > 0:
                nxt.append(self.eng.fetch_beliefs())
            i0 += n
        self.chunks = nxt
        if sum(c.shape[0] for c in nxt) <= 65535 and nxt:       # few enough simulations left: stay on the device from here on
            self.eng.set_beliefs(np.concatenate(nxt))
            self.chunks = None


class Agent:
    """Greedy agent over a value function (``src/pomdp.py:2948-3380``): best action for a belief, single
    simulations, and n simulations advanced together with the belief block on the host or in the HIP engine."""

    def __init__(self, model: Model, value_function: Union[ValueFunction, None] = None) -> None:
        self.model = model
        self.value_function = value_function

    def train(self, solver: PBVI_Solver, expansions: int, horizon: int) -> SolverHistory:
        self.value_function, hist = solver.solve(self.model, expansions, horizon)
        return hist

    def get_best_action(self, belief):
        """``actions[argmax_v b.alpha_v]`` for one ``Belief`` (returns int) or a ``[n,S]`` array (``:3005-3034``)."""
        vf = self.value_function
        assert vf is not None, "No value function, training probably has to be run..."
        single = isinstance(belief, Belief)
        arr = belief.values[None, :] if single else belief
        if vf.is_on_gpu:
            eng = vf.model.engine
            eng.sync_rows('alpha', vf.alpha_vector_list, lambda v: v.values)
            parts = []
            for i0 in range(0, arr.shape[0], _DeviceBeliefBlock.CHUNK):      # the engine takes 65535 beliefs at a time
                eng.set_beliefs(arr[i0:i0 + _DeviceBeliefBlock.CHUNK])
                parts.append(eng.max_value_resident()[1])
            best = parts[0] if len(parts) == 1 else np.concatenate(parts)
        else:
            best = np.argmax(np.matmul(arr, vf.alpha_vector_array.T), axis=1)
        acts = vf.actions[best]
        return int(acts[0]) if single else acts

    def simulate(self, simulator: Union[Simulation, None] = None, max_steps: int = 1000,
                 start_state: Union[int, None] = None, initial_belief: Union[Belief, None] = None,
                 print_progress: bool = True, print_stats: bool = True) -> SimulationHistory:
        assert self.value_function is not None, "No value function, training probably has to be run..."
        self.model = self.model.gpu_model if self.value_function.is_on_gpu else self.model.cpu_model
        simulator = Simulation(self.model) if simulator is None else simulator
        s = simulator.initialize_simulation(start_state=start_state)
        belief = Belief(self.model) if initial_belief is None else initial_belief
        history = SimulationHistory(self.model, start_state=s, start_belief=belief)
        t0 = datetime.now()
        for _ in range(max_steps):
            a = self.get_best_action(belief)
            r, o = simulator.run_action(a)
            belief = belief.update(a, o)
            history.add(action=a, next_state=simulator.agent_state, next_belief=belief, reward=r, observation=o)
            if simulator.is_done:
                break
        if print_stats:
            print('Simulation done:')
            print(f'\t- Runtime (s): {(datetime.now() - t0).total_seconds()}')
            print(f'\t- Steps: {len(history.states)}')
            print(f'\t- Total rewards: {sum(history.rewards)}')
            print(f'\t- End state: {self.model.state_labels[history.states[-1]]}')
        return history

    def run_n_simulations(self, simulator: Union[Simulation, None] = None, n: int = 1000, max_steps: int = 1000,
                          start_states: Union[list, int, None] = None, initial_beliefs=None,
                          reward_discount: float = 0.99, print_progress: bool = True, print_stats: bool = True):
        simulator = Simulation(self.model) if simulator is None else simulator
        assert (not isinstance(start_states, list)) or (len(start_states) == n), 'The size of the list of start states has to match n'
        assert (not isinstance(initial_beliefs, list)) or (len(initial_beliefs) == n), 'The size of the list of initial beliefs has to match n'
        t0 = datetime.now()
        histories, totals, discounted, done = [], RewardSet(), [], 0
        for i in range(n):
            h = self.simulate(simulator=simulator, max_steps=max_steps,
                              start_state=(start_states[i] if isinstance(start_states, list) else start_states),
                              initial_belief=(initial_beliefs[i] if isinstance(initial_beliefs, list) else initial_beliefs),
                              print_progress=False, print_stats=False)
            done += int(simulator.is_done)
            histories.append(h)
            totals.append(np.sum(h.rewards))
            discounted.append(h.rewards.get_total_discounted_reward(reward_discount))
        if print_stats:
            print(f'All {n} simulations done:')
            print(f'\t- Average runtime (s): {(datetime.now() - t0).total_seconds() / n}')
            print(f'\t- Simulations reached goal: {done}/{n} ({n - done} failures)')
            print(f'\t- Average step count: {sum(len(h) for h in histories) / n}')
            print(f'\t- Average total rewards: {sum(totals) / n}')
            print(f'\t- Average discounted rewards (ADR): {sum(discounted) / n}')
        return totals, histories

    def run_n_simulations_parallel(self, n: int = 1000, simulator_set: Union[SimulationSet, None] = None,
                                   max_steps: int = 1000, start_states: Union[list, int, None] = None,
                                   initial_beliefs=None, reward_discount: float = 0.99,
                                   print_progress: bool = True, print_stats: bool = True):
        """n simulations advanced in lock-step (``src/pomdp.py:3203-3380``).  Per step: best α per belief
        (GEMM + first-max), host simulator draw, Bayes update of every belief, done-filter.  With the value
        function on the GPU the belief block lives in the HIP engine for the whole run; only the ``[n]`` index,
        action and observation vectors cross the boundary each step."""
        vf = self.value_function
        assert vf is not None, "No value function, training probably has to be run..."
        on_gpu = vf.is_on_gpu
        model = self.model.gpu_model if on_gpu else self.model.cpu_model
        assert (not isinstance(start_states, list)) or (len(start_states) == n), 'The size of the list of start states has to match n'
        assert (not isinstance(initial_beliefs, list)) or (len(initial_beliefs) == n), 'The size of the list of initial beliefs has to match n'

        if initial_beliefs is None:
            b0 = np.repeat(Belief(model).values[None, :], n, axis=0)
        elif isinstance(initial_beliefs, Belief):
            b0 = np.repeat(initial_beliefs.values[None, :], n, axis=0)
        else:
            b0 = np.array([b.values for b in initial_beliefs])

        simulator_set = SimulationSet(model) if simulator_set is None else simulator_set
        start_state_array = simulator_set.initialize_simulations(n, start_states)
        block = (_DeviceBeliefBlock if on_gpu else _HostBeliefBlock)(model, vf, b0)

        done_at_step = np.full(n, -1, dtype=int)
        alive = np.arange(n)
        discount = reward_discount
        rewards_history = np.zeros((max_steps, n))
        discounted_history = np.zeros((max_steps, n))
        states_history = np.empty((max_steps + 1, n))
        states_history[0] = start_state_array
        actions_history = np.empty((max_steps, n))
        observations_history = np.empty((max_steps, n))

        t0 = datetime.now()
        for i in range(max_steps):
            best_actions = vf.actions[block.best_vectors()]
            rewards, observations = simulator_set.run_actions(best_actions)
            finished = simulator_set.is_done
            block.advance(best_actions, observations, ~finished)

            rewards_history[i, alive] = rewards
            discounted_history[i, alive] = rewards * discount
            states_history[i + 1, alive] = simulator_set.agent_states
            actions_history[i, alive] = best_actions
            observations_history[i, alive] = observations
            done_at_step[alive[finished]] = i

            alive = alive[~finished]
            simulator_set.n = len(alive)
            simulator_set.agent_states = simulator_set.agent_states[~finished]
            simulator_set.simulations = simulator_set.simulations[~finished]
            simulator_set.is_done = finished[~finished]
            discount *= reward_discount
            if len(alive) == 0:
                break

        histories, steps_sum = [], 0
        b_start = Belief(model)
        for i, s0 in enumerate(start_state_array):
            h = SimulationHistory(self.model, int(s0), b_start)
            last = int(done_at_step[i]) if done_at_step[i] >= 0 else max_steps
            steps_sum += last
            h.states = states_history[:last + 1, i].tolist()
            h.actions = actions_history[:last, i].tolist()
            h.observations = observations_history[:last, i].tolist()
            h.rewards = rewards_history[:last, i].tolist()
            histories.append(h)
        n_done = int(np.sum(done_at_step >= 0))
        if print_stats:
            print(f'All {n} simulations done in {(datetime.now() - t0).total_seconds():.3f}s:')
            print(f'\t- Simulations reached goal: {n_done}/{n} ({n - n_done} failures)')
            print(f'\t- Average step count: {steps_sum / n}')
            print(f'\t- Average total rewards: {np.sum(rewards_history) / n}')
            print(f'\t- Average discounted rewards (ADR): {np.sum(discounted_history) / n}')
        return RewardSet(np.sum(rewards_history, axis=0).tolist()), histories


# --------------------------------------------------------------------------- #
# Cassandra .POMDP files (inputs of BASELINE configs 0-1)
# --------------------------------------------------------------------------- #
def load_POMDP_file(file_name: str) -> Tuple[Model, PBVI_Solver]:
    """Parse a Cassandra ``.POMDP`` file into ``(Model, PBVI_Solver(gamma))``.

    Covers the constructs the reference's loader handles (``src/pomdp.py:3383-3737``):
    ``discount/values/states/actions/observations/start`` headers and ``T``, ``O``,
    ``R`` entries given as single values, rows, or matrices with ``uniform`` /
    ``identity`` shortcuts and ``*`` wildcards.  Rewards land in ``R[s,a,s',o]``.

    Bit-identical tables to the reference's loader on 20 of the 25 example models it ships
    (``tests/golden/pomdp_file_tables.json``).  The other five: three it rejects itself; ``hanks.95`` and
    ``network.95`` write fully specified entries with the value on the next line (``T: a : s : s'`` / ``0.1``),
    which the reference reads as a row assignment (its tables stop being stochastic) and this loader reads as the
    file format defines it, one entry.
    """
    with open(file_name) as fh:
        lines = [ln.split('#')[0].strip() for ln in fh]
    lines = [ln for ln in lines if ln]

    gamma = 1.0
    names = {'states': None, 'actions': None, 'observations': None}
    start = None
    T = Ob = Rw = None

    def counts():
        return len(names['states']), len(names['actions']), len(names['observations'])

    def ensure_tables():
        nonlocal T, Ob, Rw
        if T is None and all(v is not None for v in names.values()):
            S, A, O = counts()
            T = np.zeros((S, A, S))
            Ob = np.zeros((S, A, O))
            Rw = np.zeros((S, A, S, O))

    def sel(kind: str, tok: str):
        n = len(names[kind])
        if tok == '*':
            return list(range(n))
        if tok.isnumeric():
            return [int(tok)]
        return [names[kind].index(tok)]

    def is_number(tok: str) -> bool:
        try:
            float(tok)
            return True
        except ValueError:
            return False

    i = 0
    while i < len(lines):
        ln = lines[i]
        i += 1
        head, _, rest = ln.partition(':')
        head = head.strip()
        if head == 'discount':
            gamma = float(rest)
        elif head == 'values':
            pass
        elif head in names:
            toks = rest.split()
            if len(toks) == 1 and toks[0].isnumeric():
                prefix = {'states': 's', 'actions': 'a', 'observations': 'o'}[head]
                names[head] = [f'{prefix}{k}' for k in range(int(toks[0]))]
            else:
                names[head] = toks
            ensure_tables()
        elif head == 'start':
            toks = rest.split()
            if not toks:
                toks = lines[i].split()
                i += 1
            assert len(toks) == len(names['states']), 'Not enough states in initial belief'
            start = np.array([float(t) for t in toks])
        elif head in ('T', 'O', 'R'):
            S, A, O = counts()
            fields = [f.strip() for f in rest.split(':')]
            value = None
            last = fields[-1].split()
            if len(last) > 1:                       # "... : x value"
                fields[-1] = last[0]
                value = float(last[1])
            keys = [f for f in fields if f != '']
            full = {'T': 3, 'O': 3, 'R': 4}[head]
            if value is None and len(keys) == full + 1 and is_number(keys[-1]):
                value = float(keys.pop())
            if value is None and len(keys) == full:   # fully specified entry, its single value on the next line
                value = float(lines[i].split()[0])
                i += 1
            acts = sel('actions', keys[0])
            if head == 'T':
                if len(keys) == 3:
                    for a in acts:
                        for s in sel('states', keys[1]):
                            for sp in sel('states', keys[2]):
                                T[s, a, sp] = value
                elif len(keys) == 2:
                    row = lines[i].split()
                    i += 1
                    for a in acts:
                        for s in sel('states', keys[1]):
                            T[s, a, :] = (np.ones(S) / S) if row[0] == 'uniform' else [float(t) for t in row]
                else:
                    first = lines[i].split()
                    if first[0] in ('uniform', 'identity'):
                        i += 1
                        for a in acts:
                            T[:, a, :] = (np.ones((S, S)) / S) if first[0] == 'uniform' else np.eye(S)
                    else:
                        mat = np.array([[float(t) for t in lines[i + k].split()] for k in range(S)])
                        i += S
                        for a in acts:
                            T[:, a, :] = mat
            elif head == 'O':
                if len(keys) == 3:
                    for a in acts:
                        for sp in sel('states', keys[1]):
                            for o in sel('observations', keys[2]):
                                Ob[sp, a, o] = value
                elif len(keys) == 2:
                    row = lines[i].split()
                    i += 1
                    for a in acts:
                        for sp in sel('states', keys[1]):
                            Ob[sp, a, :] = (np.ones(O) / O) if row[0] == 'uniform' else [float(t) for t in row]
                else:
                    first = lines[i].split()
                    if first[0] == 'uniform':
                        i += 1
                        for a in acts:
                            Ob[:, a, :] = np.ones((S, O)) / O
                    else:
                        mat = np.array([[float(t) for t in lines[i + k].split()] for k in range(S)])
                        i += S
                        for a in acts:
                            Ob[:, a, :] = mat
            else:  # R
                if len(keys) == 4:
                    for a in acts:
                        for s in sel('states', keys[1]):
                            for sp in sel('states', keys[2]):
                                for o in sel('observations', keys[3]):
                                    Rw[s, a, sp, o] = value
                elif len(keys) == 3:
                    row = [float(t) for t in lines[i].split()]
                    i += 1
                    for a in acts:
                        for s in sel('states', keys[1]):
                            for sp in sel('states', keys[2]):
                                Rw[s, a, sp, :] = row
                elif len(keys) == 2:
                    mat = np.array([[float(t) for t in lines[i + k].split()] for k in range(S)])
                    i += S
                    for a in acts:
                        for s in sel('states', keys[1]):
                            Rw[s, a, :, :] = mat
                else:
                    raise Exception('Need more than 1 parameter for rewards')

    params = dict(states=names['states'], actions=names['actions'], observations=names['observations'],
                  transitions=T, observation_table=Ob, rewards=Rw)
    if start is not None:
        params['start_probabilities'] = start
    return Model(**params), PBVI_Solver(gamma=gamma)
