"""Host-side mirror of the reference's ``src/mdp.py`` for the backup path.

Only what the PBVI backup path and its callers need is here: ``log``, the MDP
``Model`` (reachable-state tables), ``AlphaVector`` / ``ValueFunction`` (the
backup's output contract: byte-exact dedup, new-then-old union, level-2
domination prune), ``VI_Solver`` (seeds FSVI/HSVI) and the simulation containers
(``RewardSet``, ``SimulationHistory``, ``Simulation``, ``Agent``) that policy
evaluation builds on.  Plotting, videos and file
persistence of the reference are out of scope.

Device residency replaces the reference's CuPy twins (``src/mdp.py:533-590``,
``:782-831``): ``Model.gpu_model`` returns a twin bound to a HIP engine handle
(``engine.Engine``); arrays stay NumPy on the host and the engine owns the
device copies.  If the HIP library cannot be loaded, asking for the GPU twin
raises -- there is no silent CPU fallback.
"""
from __future__ import annotations

import os
import pickle
import random
from datetime import datetime
from typing import Tuple, Union

import numpy as np

gpu_support = True   # the HIP engine is always the GPU backend; loading errors surface on first use


def log(content: str) -> None:
    """Timestamped print (``src/mdp.py:40-49``)."""
    print(f'[{datetime.now().strftime("%m/%d/%Y, %H:%M:%S")}] ' + content)


_QUIET = [False]


def set_quiet(q: bool = True) -> None:
    """Silence model-construction logging (tests / bench)."""
    _QUIET[0] = bool(q)


def _log(msg: str) -> None:
    if not _QUIET[0]:
        log(msg)


class Model:
    """MDP model with the reachable-state (padded-ELL) transition representation.

    Mirrors ``src/mdp.py:52-408``: same constructor arguments and attributes
    (``states, actions, transition_table, reachable_states,
    reachable_probabilities, reachable_state_count, expected_rewards_table,
    start_probabilities, end_states, end_actions, is_on_gpu, gpu_model,
    cpu_model``).
    """

    def __init__(self, states, actions, transitions=None, reachable_states=None, rewards=None,
                 rewards_are_probabilistic: bool = False, state_grid=None, start_probabilities=None,
                 end_states: list = [], end_actions: list = []):
        self._alt_model = None
        self.is_on_gpu = False
        self._engine = None
        self._engine_dtype = None

        # states
        self.state_grid = None
        if isinstance(states, int):
            self.state_labels = [f's_{i}' for i in range(states)]
        elif isinstance(states, list) and states and all(isinstance(r, list) for r in states):
            width = len(states[0])
            assert all(len(r) == width for r in states), "All sublists of states must be of equal size"
            self.state_labels = [lab for r in states for lab in r]
            self.state_grid = np.arange(len(states) * width).reshape(len(states), width)
        else:
            self.state_labels = [lab for lab in states if isinstance(lab, str)]
        self.state_count = len(self.state_labels)
        self.states = np.arange(self.state_count)

        # actions
        self.action_labels = [f'a_{i}' for i in range(actions)] if isinstance(actions, int) else actions
        self.action_count = len(self.action_labels)
        self.actions = np.arange(self.action_count)
        _log(f'MDP model: {self.state_count} states, {self.action_count} actions')

        S, A = self.state_count, self.action_count

        # transitions
        self.reachable_states = None
        if reachable_states is not None:
            self.reachable_states = np.array(reachable_states)
            assert self.reachable_states.shape[:2] == (S, A), \
                f"Reachable states provided is not of the expected shape (received {self.reachable_states.shape}, expected ({S}, {A}, :))"
            self.reachable_state_count = self.reachable_states.shape[2]

        self.transition_table = None
        self.transition_function = None
        if transitions is None:
            if reachable_states is None:
                rnd = np.random.rand(S, A, S)
                self.transition_table = rnd / np.sum(rnd, axis=2, keepdims=True)
        elif callable(transitions):
            self.transition_function = transitions
            try:
                self.transition_table = np.fromfunction(transitions, (S, A, S))
            except MemoryError:
                _log('    > [Warning] Not enough memory to store transition table, using transition function provided...')
        else:
            self.transition_table = np.array(transitions)
            assert self.transition_table.shape == (S, A, S), \
                f"Transitions table provided doesnt have the right shape, it should be SxAxS (expected {(S, A, S)}, received {self.transition_table.shape})"

        self.rewards_are_probabilistic = rewards_are_probabilistic

        # grid
        if state_grid is not None:
            self.state_grid = np.array(state_grid)
        elif self.state_grid is None:
            self.state_grid = np.arange(S).reshape((1, S))

        # start distribution
        if start_probabilities is not None:
            assert len(start_probabilities) == S
            self.start_probabilities = np.array(start_probabilities, dtype=float)
        else:
            self.start_probabilities = np.full(S, 1 / S)

        self.end_states = end_states
        self.end_actions = end_actions

        # reachable states derived from the dense table (pad with unused low indices, prob 0)
        if self.reachable_states is None:
            per_pair = []
            for s in range(S):
                row = []
                for a in range(A):
                    if self.transition_table is not None:
                        row.append(np.flatnonzero(self.transition_table[s, a, :] > 0).tolist())
                    else:
                        row.append([sn for sn in range(S) if self.transition_function(s, a, sn) > 0])
                per_pair.append(row)
            self.reachable_state_count = max(len(l) for row in per_pair for l in row)
            for row in per_pair:
                for l in row:
                    filler = 0
                    while len(l) < self.reachable_state_count:
                        if filler not in l:
                            l.append(filler)
                        filler += 1
            self.reachable_states = np.array(per_pair, dtype=int)
        _log(f'- at most {self.reachable_state_count} reachable states per state-action pair')

        if self.transition_table is not None:
            self.reachable_probabilities = self.transition_table[self.states[:, None, None],
                                                                 self.actions[None, :, None],
                                                                 self.reachable_states]
        elif self.transition_function is not None:
            rs = self.reachable_states
            self.reachable_probabilities = np.fromfunction(
                lambda s, a, r: self.transition_function(s.astype(int), a.astype(int),
                                                         rs[s.astype(int), a.astype(int), r.astype(int)]), rs.shape)
        else:
            self.reachable_probabilities = np.full(self.reachable_states.shape, 1 / self.reachable_state_count)

        # rewards (skipped when the POMDP subclass defines them: rewards == -1)
        self.immediate_reward_table = None
        self.immediate_reward_function = None
        self._min_reward = None
        self._max_reward = None
        self.expected_rewards_table = None
        if isinstance(rewards, int) and rewards == -1:
            return
        if rewards is None:
            if len(self.end_states) > 0 or len(self.end_actions) > 0:
                self.immediate_reward_function = self._end_reward_function
            else:
                self.immediate_reward_table = np.random.rand(S, A, S)
        elif callable(rewards):
            self.immediate_reward_function = rewards
        else:
            self.immediate_reward_table = np.array(rewards)
            assert self.immediate_reward_table.shape == (S, A, S), "Rewards table doesnt have the right shape, it should be SxAxS"
        if self.immediate_reward_table is not None:
            reach_r = self.immediate_reward_table[self.states[:, None, None], self.actions[None, :, None], self.reachable_states]
        else:
            rs = self.reachable_states
            reach_r = np.fromfunction(lambda s, a, r: self.immediate_reward_function(
                s.astype(int), a.astype(int), rs[s.astype(int), a.astype(int), r.astype(int)]), rs.shape)
        self._min_reward = float(np.min(reach_r))
        self._max_reward = float(np.max(reach_r))
        self.expected_rewards_table = np.einsum('sar,sar->sa', self.reachable_probabilities, reach_r)

    def _end_reward_function(self, s, a, sn):
        return (np.isin(sn, self.end_states) | np.isin(a, self.end_actions)).astype(int)

    def transition(self, s: int, a: int) -> int:
        """Sample a successor state (``src/mdp.py:415-438``)."""
        if self.reachable_state_count == 1:
            return int(self.reachable_states[s, a, 0])
        return int(self.reachable_states[s, a, _draw_index(self.reachable_probabilities[s, a])])

    def reward(self, s: int, a: int, s_p: int):
        """Reward of landing in ``s_p`` after ``a`` in ``s``; a Bernoulli draw when rewards are
        probabilities (``src/mdp.py:441-465``)."""
        if self.immediate_reward_table is not None:
            r = float(self.immediate_reward_table[s, a, s_p])
        else:
            r = float(self.immediate_reward_function(s, a, s_p))
        if self.rewards_are_probabilistic:
            return 1 if random.random() < r else 0
        return r

    def get_coords(self, item):
        """Grid position(s) of a state id or a list of ids on ``state_grid`` (``src/mdp.py:467-484``)."""
        ids = [item] if isinstance(item, int) else item
        coords = [np.argwhere(self.cpu_model.state_grid == s)[0] for s in ids]
        return coords[0] if isinstance(item, int) else coords

    # -- persistence (src/mdp.py:487-530): the host model as a pickle ------- #
    def __getstate__(self):
        state = dict(self.__dict__)
        state['_alt_model'] = None           # the GPU twin and its engine handle are rebuilt on demand
        state['_engine'] = None
        state['_engine_dtype'] = None
        state['is_on_gpu'] = False
        return state

    def save(self, file_name: str, path: str = './Models') -> None:
        if not os.path.exists(path):
            print('Folder does not exist yet, creating it...')
            os.makedirs(path)
        if not file_name.endswith('.pck'):
            file_name += '.pck'
        with open(path + '/' + file_name, 'wb') as fh:
            pickle.dump(self.cpu_model, fh)

    @classmethod
    def load_from_file(cls, file: str) -> 'Model':
        """Load a model written by ``save``.  Only for files you wrote yourself: a pickle runs code when loaded."""
        with open(file, 'rb') as fh:
            return pickle.load(fh)

    # -- residency ------------------------------------------------------- #
    def to_gpu(self, dtype: str = 'f64', device: int = None) -> 'Model':
        """GPU twin bound to a HIP engine of the given arithmetic type.  ``device`` defaults to ``LOCAL_RANK`` (one
        process per GPU under ``torch.distributed.run``; the reference's drivers pick the card with
        ``cupy.cuda.runtime.setDevice``, ``Experiments/Olfactory Navigation/run_test.py:12``), else 0."""
        if self.is_on_gpu:
            return self
        if device is None:
            import os
            device = int(os.environ.get('LOCAL_RANK', 0))
            if device > 0:
                from .engine import device_count
                device %= max(device_count(), 1)        # more ranks than cards (a rehearsal on a smaller box): share them
        if self._alt_model is None or self._alt_model._engine_dtype != dtype:
            from .engine import Engine          # raises if the HIP library is missing
            twin = object.__new__(self.__class__)
            twin.__dict__.update(self.__dict__)
            twin.is_on_gpu = True
            twin._alt_model = self
            twin._engine_dtype = dtype
            twin._engine = Engine.for_model(self, dtype=dtype, device=device)
            self._alt_model = twin
        return self._alt_model

    @property
    def gpu_model(self) -> 'Model':
        return self.to_gpu(self._alt_model._engine_dtype if (self._alt_model is not None and not self.is_on_gpu) else 'f64')

    @property
    def cpu_model(self) -> 'Model':
        return self._alt_model if self.is_on_gpu else self

    @property
    def engine(self):
        assert self.is_on_gpu, "model is not on the GPU; use model.gpu_model"
        return self._engine


def _draw_index(p) -> int:
    """Index drawn with probabilities ``p``: what ``int(np.random.choice(len(p), size=1, p=p)[0])`` returns, consuming the
    same one double of NumPy's global stream -- the legacy generator's own statements (``cdf = p.cumsum(); cdf /= cdf[-1];
    cdf.searchsorted(random_sample(1), side='right')``) without its per-call argument checks (13.5 -> 3 us; an FSVI
    expansion draws 99 observations one at a time).  The reference's trajectories depend on that stream, so the draw has
    to stay this one, call for call."""
    cdf = np.cumsum(p, dtype=np.float64)
    # np.random.choice's own argument checks, at the cost of one comparison and one min (a denormalised belief or
    # observation row must raise here as it does in the reference, not be renormalised silently)
    if not abs(cdf[-1] - 1.0) <= 1.4901161193847656e-08:          # sqrt(eps of float64), choice's tolerance
        raise ValueError('probabilities do not sum to 1')
    if np.min(p) < 0:
        raise ValueError('probabilities are not non-negative')
    cdf /= cdf[-1]
    return int(cdf.searchsorted(np.random.random_sample(), side='right'))


_HASH_WEIGHTS = {}


def _same_bytes(a, b) -> bool:
    """``a.tobytes() == b.tobytes()`` without materialising the two byte strings (240 KB each at S = 30000: the dedup
    dictionaries of a solve compare ~17 000 pairs of rows with equal hashes, nearly all of them true repeats)."""
    if a is b:
        return True
    if a.dtype != b.dtype or a.shape != b.shape:
        return a.tobytes() == b.tobytes()
    if a.dtype.itemsize in (4, 8) and a.flags.c_contiguous and b.flags.c_contiguous:
        t = np.uint32 if a.dtype.itemsize == 4 else np.uint64
        return bool(np.array_equal(a.view(t), b.view(t)))       # bit patterns: -0.0 != 0.0 and NaN == NaN, like bytes
    return a.tobytes() == b.tobytes()


def _row_hash(row) -> int:
    """``sum_i bits_i * (2 i + 1) mod 2^64`` over the fp32 / fp64 bit patterns of a 1-D row: the number the dedup keys of
    the belief and alpha-vector containers hash by.  The HIP engine computes the same number for rows it produced
    (``k_row_hash``: ``pbvi_backup_fetch_row_hashes``, ``pbvi_belief_walk_keys``).  Position-weighted, so shifted copies of
    a row do not collide."""
    a = np.ascontiguousarray(row)
    if a.dtype.itemsize not in (4, 8) or a.ndim != 1:
        return hash(a.tobytes()) & 0xFFFFFFFFFFFFFFFF
    bits = a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint64)
    w = _HASH_WEIGHTS.get(bits.shape[0])
    if w is None:
        w = _HASH_WEIGHTS[bits.shape[0]] = np.arange(bits.shape[0], dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    with np.errstate(over='ignore'):
        return int(np.dot(bits.astype(np.uint64, copy=False), w))          # integer dot: wraps modulo 2^64 like the device sum


class _RowKey(int):
    """Dictionary key with the semantics of the row's bytes (the reference keys its dedup dictionaries on
    ``values.tobytes()``, ``src/mdp.py:660-669``, ``src/pomdp.py:562-571``) without copying and hashing a quarter of
    a megabyte per row: the key IS an integer -- ``_row_hash``, one single-threaded pass, equal bytes give equal numbers --
    so dictionaries hash it at C speed, and equality compares the bytes, which only happens when two keys carry the same
    number, i.e. practically only for equal rows.  (A plain sum of the bit patterns was used first: a walk's successive
    beliefs and a solve's alpha-vectors are largely shifted copies of one another on a wrap-around grid, so sums collide --
    8 000 byte comparisons of 240 KB rows in a 300-expansion solve; the position weights end that.)  Used by the belief
    containers; ``_AlphaKey`` is the same for alpha-vector rows."""

    def __new__(cls, row):
        if isinstance(row, int):                            # copy / pickle rebuild: the row comes back through __dict__
            return int.__new__(cls, row)
        h = _row_hash(row)
        self = int.__new__(cls, h)
        self.row = row
        return self

    @classmethod
    def from_sum(cls, bit_sum: int, row) -> '_RowKey':
        """Key of ``row`` whose ``_row_hash`` is already known (the engine computes it for rows it produced)."""
        self = int.__new__(cls, bit_sum)
        self.row = row
        return self

    __hash__ = int.__hash__

    def __eq__(self, other) -> bool:
        return int.__eq__(self, other) is True and _same_bytes(self.row, other.row)

    def __ne__(self, other) -> bool:
        return not self.__eq__(other)


class _AlphaKey(int):
    """Dictionary key of an alpha-vector row with the semantics of the row's bytes (the reference keys
    ``ValueFunction._uniqueness_dict`` on ``values.tobytes()``, ``src/mdp.py:660-669``): the key IS an integer -- the
    position-weighted sum ``sum_i bits_i * (2 i + 1) mod 2^64`` of the row's fp32 / fp64 bit patterns, so equal bytes give
    equal keys and shifted copies of a row (a solve's alpha-vectors on a wrap-around grid) do not collide as they do under
    the plain sum of ``_RowKey`` -- and equality compares the bytes, which only happens when two keys carry the same number,
    i.e. practically only for equal rows.  For rows the engine produced the number comes from the device
    (``pbvi_backup_fetch_row_hashes``); here it costs one pass instead of a 120-240 KB copy plus a byte hash."""

    def __new__(cls, row):
        if isinstance(row, int):                            # copy / pickle rebuild: the row comes back through __dict__
            return int.__new__(cls, row)
        self = int.__new__(cls, _row_hash(row))
        self.row = row
        return self

    hash_of = staticmethod(_row_hash)

    @classmethod
    def from_hash(cls, h: int, row) -> '_AlphaKey':
        """Key of ``row`` whose hash is already known (the engine computes it for rows it produced)."""
        self = int.__new__(cls, int(h))
        self.row = row
        return self

    __hash__ = int.__hash__

    def __eq__(self, other) -> bool:
        return int.__eq__(self, other) is True and _same_bytes(self.row, other.row)

    def __ne__(self, other) -> bool:
        return not self.__eq__(other)


class AlphaVector:
    """One hyperplane over the state space and its action (``src/mdp.py:593-608``)."""

    def __init__(self, values: np.ndarray, action: int) -> None:
        self.values = values
        self.action = int(action)

    @property
    def key(self) -> _AlphaKey:
        """Dedup key of the row as it is NOW (the reference computes ``values.tobytes()`` whenever a container is built).
        Rows the engine produced carry the device's hash (``_hash``, set with the row, which nothing in this package
        mutates); anything else is hashed here."""
        h = self.__dict__.get('_hash')
        return _AlphaKey.from_hash(h, self.values) if h is not None else _AlphaKey(self.values)


class ValueFunction:
    """A set of alpha-vectors with the reference's container semantics.

    * constructor dedup keyed on the row's exact bytes -- first position, last
      action wins (``src/mdp.py:660-669``);
    * ``extend``: this set's vectors first, then the other's; on identical bytes
      the other's object replaces ours in place (``src/mdp.py:763-779``);
    * ``prune(level=2)``: drop every row some other row dominates point-wise
      (``src/mdp.py:857-866``); runs on the HIP engine when the set is on the GPU.
    """

    def __init__(self, model: Model, alpha_vectors: Union[list, np.ndarray] = [], action_list=[]):
        self.model = model
        self._vector_array = None
        self._actions = None
        self.is_on_gpu = bool(getattr(model, 'is_on_gpu', False))
        if isinstance(alpha_vectors, list):
            assert all(v.values.shape[0] == model.state_count for v in alpha_vectors), \
                f"Some or all alpha vectors in the list provided dont have the right size, they should be of shape: {model.state_count}"
            vectors = alpha_vectors
        else:
            expected = (len(action_list), model.state_count)
            assert alpha_vectors.shape == expected, \
                f"Alpha vector array does not have the right shape (received: {alpha_vectors.shape}; expected: {expected})"
            vectors = [AlphaVector(row, act) for row, act in zip(alpha_vectors, action_list)]
        self._uniqueness_dict = {v.key: v for v in vectors}
        self._vector_list = list(self._uniqueness_dict.values())
        self._pruning_level = 1

    @property
    def alpha_vector_list(self) -> list:
        if self._vector_list is None:
            self._vector_list = [AlphaVector(r, a) for r, a in zip(self._vector_array, self._actions)]
        return self._vector_list

    def _materialise(self) -> None:
        if self._vector_array is None:
            if len(self._vector_list) == 0:
                self._vector_array = np.zeros((0, self.model.state_count))
            else:
                self._vector_array = np.array([v.values for v in self._vector_list])

    @property
    def alpha_vector_array(self) -> np.ndarray:
        self._materialise()
        return self._vector_array

    @property
    def actions(self) -> np.ndarray:
        # on its own: asking for the actions (the simulators do, every step) must not re-stack the whole V x S matrix
        if self._actions is None:
            self._actions = np.array([v.action for v in self.alpha_vector_list], dtype=int)
        return self._actions

    def __len__(self) -> int:
        return len(self._vector_list) if self._vector_list is not None else self._vector_array.shape[0]

    def __add__(self, other: 'ValueFunction') -> 'ValueFunction':
        out = object.__new__(self.__class__)
        out.model = self.model
        out.is_on_gpu = self.is_on_gpu
        out._uniqueness_dict = {**self._uniqueness_dict, **other._uniqueness_dict}
        out._vector_list = list(out._uniqueness_dict.values())
        out._vector_array = None
        out._actions = None
        out._pruning_level = 1
        return out

    def append(self, alpha_vector: AlphaVector) -> None:
        assert alpha_vector.values.shape[0] == self.model.state_count, "Vector to add to value function doesn't have the right size"
        self._uniqueness_dict[alpha_vector.key] = alpha_vector
        self._vector_list = list(self._uniqueness_dict.values())
        self._dev_ids = None
        self._vector_array = None
        self._actions = None

    def extend(self, other: 'ValueFunction') -> None:
        mine, theirs = self._uniqueness_dict, other._uniqueness_dict
        # device store ids of the result (Engine.row_ids caches them per container): this set's rows keep their
        # slots, then the other's rows that are new.  A slot whose bytes the other set has too is taken over by the
        # other's object (dict.update) AND by its store id: the two store rows are equal, and with the older id the id set
        # of a solve's value function only ever grows -- Engine.max_value_objects extends a cached maximum over a SUBSET
        # of ids; with the newer id every such repeat sent compute_change back to scoring all beliefs x all rows
        # (two 0.1 s GEMMs in the 300-expansion run).
        carried = None
        a, b = getattr(self, '_dev_ids', None), getattr(other, '_dev_ids', None)
        if a is None and len(mine) and all(hasattr(v, '_dev') for v in mine.values()):
            tag = next(iter(mine.values()))._dev[0]
            if all(v._dev[0] == tag for v in mine.values()):
                a = (tag, np.fromiter((v._dev[1] for v in mine.values()), dtype=np.int32, count=len(mine)))
        if a is not None and b is not None and a[0] == b[0] and len(a[1]) == len(mine) and len(b[1]) == len(theirs):
            fresh = np.fromiter((k not in mine for k in theirs), dtype=bool, count=len(theirs))
            ids_mine = a[1]
            if not fresh.all():
                slot_of = {k: i for i, k in enumerate(mine)}
                ids_mine = ids_mine.copy()
                for j, k in enumerate(theirs):
                    if not fresh[j]:
                        ids_mine[slot_of[k]] = b[1][j]
            carried = (a[0], np.concatenate([ids_mine, b[1][fresh]]))
        mine.update(theirs)
        self._vector_list = list(mine.values())
        self._dev_ids = carried
        self._vector_array = None
        self._actions = None
        self._pruning_level = 1

    def to_gpu(self) -> 'ValueFunction':
        gm = self.model.gpu_model
        return ValueFunction(gm, [AlphaVector(v.values, v.action) for v in self.alpha_vector_list])

    def to_cpu(self) -> 'ValueFunction':
        cm = self.model.cpu_model
        return ValueFunction(cm, [AlphaVector(np.asarray(v.values, dtype=np.float64), v.action) for v in self.alpha_vector_list])

    # -- on-disk formats (src/mdp.py:909-1036): one row per alpha-vector, columns ``action, <state labels>`` -- #
    def _frame(self, path: str):
        import pandas as pd
        if not os.path.exists(path):
            print('Folder does not exist yet, creating it...')
            os.makedirs(path)
        data = np.concatenate((np.asarray(self.actions)[:, None], self.alpha_vector_array), axis=1)
        return pd.DataFrame(data, columns=['action', *self.model.state_labels])

    @staticmethod
    def _file_name(file_name: Union[str, None], ext: str) -> str:
        if file_name is None:
            file_name = datetime.now().strftime('%Y%m%d_%H%M%S') + '_value_function' + ext
        return file_name if ext in file_name else file_name + ext

    def save(self, path: str = './ValueFunctions', file_name: Union[str, None] = None, compress: bool = False) -> None:
        """CSV (optionally gzip, suffix ``.gzip``) in the reference's layout (``src/mdp.py:930-964``)."""
        df = self._frame(path)
        name = self._file_name(file_name, '.csv')
        if compress:
            name += '.gzip'
        df.to_csv(path + '/' + name, index=False, compression='gzip' if compress else None)

    def save_parquet(self, path: str = './ValueFunctions', file_name: Union[str, None] = None) -> None:
        """Parquet in the reference's layout (``src/mdp.py:967-990``)."""
        self._frame(path).to_parquet(path + '/' + self._file_name(file_name, '.parquet'), index=False)

    @classmethod
    def load_from_file(cls, file: str, model: Model) -> 'ValueFunction':
        """Read a CSV (gzip when the name contains ``.gzip``) written by ``save`` or by the reference
        (``src/mdp.py:993-1013``)."""
        import pandas as pd
        rows = pd.read_csv(file, header=0, index_col=False, compression='gzip' if '.gzip' in file else None).to_numpy()
        return cls(model, alpha_vectors=rows[:, 1:], action_list=rows[:, 0].astype(int))

    @classmethod
    def load_from_parquet(cls, file: str, model: Model) -> 'ValueFunction':
        """Read a parquet file written by ``save_parquet`` or by the reference (``src/mdp.py:1016-1036``)."""
        import pandas as pd
        rows = pd.read_parquet(file).to_numpy()
        return cls(model, alpha_vectors=rows[:, 1:], action_list=rows[:, 0].astype(int))

    def prune(self, level: int = 1) -> None:
        if level < self._pruning_level or level > 3:
            log("Attempting to prune a value function to a level already reached. Returning 'self'")
            return
        if level >= 3:
            raise NotImplementedError("LP pruning (level 3) is broken in the reference (src/mdp.py:872) and not provided")
        if level >= 2 and self._pruning_level < 2:
            if self.is_on_gpu:      # on the objects: kept vectors keep their device-store rows, nothing is re-stacked
                vecs = self.alpha_vector_list
                keep = self.model.engine.prune_dominated_objects(vecs, lambda v: v.values, owner=self)
                items = list(self._uniqueness_dict.items())
                if len(items) == len(vecs):
                    self._uniqueness_dict = {k: v for (k, v), kp in zip(items, keep) if kp}
                else:                                        # list holds duplicates the dictionary folded: rebuild
                    self._uniqueness_dict = {v.key: v for v, kp in zip(vecs, keep) if kp}
                self._vector_list = list(self._uniqueness_dict.values())
                self._vector_array = None
                self._actions = None
            else:
                arr = self.alpha_vector_array
                keep = np.zeros(arr.shape[0], dtype=bool)
                for i, v in enumerate(arr):
                    keep[i] = np.count_nonzero(np.all(arr >= v, axis=1)) == 1
                self._vector_array = arr[keep]
                self._actions = self.actions[keep]
                self._uniqueness_dict = {_AlphaKey(r): AlphaVector(r, a) for r, a in zip(self._vector_array, self._actions)}
                self._vector_list = list(self._uniqueness_dict.values())
            self._dev_ids = None
        self._pruning_level = level


class SolverHistory:
    """Timing / size bookkeeping of a value-iteration run (subset of ``src/mdp.py:1281-1400``)."""

    def __init__(self, tracking_level: int, model: Model, gamma: float, eps: float, initial_value_function=None):
        self.tracking_level = tracking_level
        self.model = model
        self.gamma = gamma
        self.eps = eps
        self.run_ts = datetime.now()
        self.iteration_times = []
        self.value_function_changes = []
        self.value_functions = [initial_value_function] if tracking_level >= 2 else []

    def add(self, iteration_time: float, value_function_change: float, value_function) -> None:
        if self.tracking_level >= 1:
            self.iteration_times.append(float(iteration_time))
            self.value_function_changes.append(float(value_function_change))
        if self.tracking_level >= 2:
            self.value_functions.append(value_function)

    @property
    def solution(self):
        assert self.tracking_level >= 2, "Tracking level is set too low, increase it to 2 if you want to have value function tracking as well."
        return self.value_functions[-1]

    @property
    def summary(self) -> str:
        return (f'Summary of Value Iteration run\n  - Model: {self.model.state_count}-state, {self.model.action_count}-action\n'
                f'  - Converged or stopped after {len(self.iteration_times)} iterations and {sum(self.iteration_times):.4f}s\n')


class Solver:
    def solve(self, model):
        raise Exception("Method has to be implemented by subclass...")


class VI_Solver(Solver):
    """MDP value iteration (``src/mdp.py:1414-1525``); seeds FSVI/HSVI.  ``use_gpu=True`` runs the sweeps on the
    device (``pbvi_mdp_value_iteration``): the whole loop is enqueued in batches and only the per-sweep change
    values come back until convergence."""

    def __init__(self, horizon: int = 10000, gamma: float = 0.99, eps: float = 0.001):
        self.horizon = horizon
        self.gamma = gamma
        self.eps = eps

    def solve(self, model: Model, initial_value_function=None, use_gpu: bool = False,
              history_tracking_level: int = 1, print_progress: bool = True):
        host = model.cpu_model
        if initial_value_function is None:
            V = ValueFunction(host, host.expected_rewards_table.T, host.actions)
        else:
            V = initial_value_function.to_cpu() if initial_value_function.is_on_gpu else initial_value_function
        v_opt = np.max(V.alpha_vector_array, axis=0)
        hist = SolverHistory(history_tracking_level, host, self.gamma, self.eps, V)
        limit = self.eps * (self.gamma / (1 - self.gamma))
        if use_gpu:
            from .engine import mdp_value_iteration          # raises if the HIP library is missing
            # per-sweep value functions (tracking level 2+) need every sweep's rows: one sweep per call then
            step = 1 if history_tracking_level >= 2 else self.horizon
            left = self.horizon
            while left > 0:
                t0 = datetime.now()
                rows, changes = mdp_value_iteration(host.reachable_states, host.reachable_probabilities,
                                                    host.expected_rewards_table, v_opt, self.gamma, limit, min(step, left))
                V = ValueFunction(host, rows, host.actions)
                v_opt = np.max(V.alpha_vector_array, axis=0)
                dt = (datetime.now() - t0).total_seconds() / max(len(changes), 1)
                for c in changes:
                    hist.add(dt, float(c), V)
                left -= len(changes)
                if len(changes) == 0 or changes[-1] < limit:
                    break
            return V, hist
        er_t = host.expected_rewards_table.T
        for _ in range(self.horizon):
            t0 = datetime.now()
            prev = v_opt
            rows = er_t + self.gamma * np.einsum('sar,sar->as', host.reachable_probabilities, v_opt[host.reachable_states])
            V = ValueFunction(host, rows, host.actions)
            v_opt = np.max(V.alpha_vector_array, axis=0)
            change = float(np.max(np.abs(v_opt - prev)))
            hist.add((datetime.now() - t0).total_seconds(), change, V)
            if change < limit:
                break
        return V, hist


# --------------------------------------------------------------------------- #
# Simulation containers (policy evaluation)
# --------------------------------------------------------------------------- #
class RewardSet(list):
    """List of step rewards (``src/mdp.py:1528-1566``; the plotting helpers are out of scope)."""

    def __init__(self, items: list = []):
        super().__init__()
        self.extend(items)

    def get_total_discounted_reward(self, gamma: float) -> float:
        """``sum_t gamma^t r_t`` (``src/mdp.py:1546-1566``)."""
        return float(np.dot(np.array(self, dtype=float), gamma ** np.arange(len(self))))


class SimulationHistory:
    """States, actions and rewards of one simulated episode (``src/mdp.py:1689-1756``)."""

    def __init__(self, model: Model, start_state: int):
        self.model = model.cpu_model
        self.states = [start_state]
        self.actions = []
        self.rewards = RewardSet()

    @property
    def grid_point_sequence(self) -> list:
        grid = self.model.state_grid
        return [[int(i[0]) for i in np.where(grid == s)] for s in self.states]

    def add(self, action: int, reward, next_state: int) -> None:
        self.actions.append(action)
        self.rewards.append(reward)
        self.states.append(next_state)

    def __len__(self) -> int:
        return len(self.states)

    # -- on-disk form (src/mdp.py:1847-1885): one row per visited state, the last row has no action / reward -- #
    def to_dataframe(self):
        import pandas as pd
        pts = self.grid_point_sequence
        return pd.DataFrame({'States': self.states,
                             'State_grid_x': [p[0] for p in pts],
                             'State_grid_y': [p[1] for p in pts],
                             'Actions': list(self.actions) + [None],
                             'Rewards': list(self.rewards) + [None]})

    @staticmethod
    def _csv_target(path: str, file_name: Union[str, None]) -> str:
        if not os.path.exists(path):
            print('Folder does not exist yet, creating it...')
            os.makedirs(path)
        if file_name is None:
            file_name = datetime.now().strftime('%Y%m%d_%H%M%S') + '_simulation.csv'
        if not file_name.endswith('.csv'):
            file_name += '.csv'
        return path + '/' + file_name

    def save(self, path: str = './Simulations', file_name: Union[str, None] = None) -> None:
        self.to_dataframe().to_csv(self._csv_target(path, file_name), index=False)


class Simulation:
    """One agent walking the model (``src/mdp.py:1888-1977``): hidden state, done flag."""

    def __init__(self, model: Model) -> None:
        self.model = model
        self.agent_state = -1
        self.is_done = True
        self.initialize_simulation()

    def initialize_simulation(self, start_state: Union[int, None] = None) -> int:
        if start_state is None:
            self.agent_state = int(self.model.states[_draw_index(self.model.start_probabilities)])
        else:
            self.agent_state = start_state
        self.is_done = False
        return self.agent_state

    def _mark_done(self, s_p: int, a: int) -> None:
        if s_p in self.model.end_states or a in self.model.end_actions:
            self.is_done = True

    def run_action(self, a: int) -> Tuple[Union[int, float], int]:
        assert not self.is_done, "Action run when simulation is done."
        s = self.agent_state
        s_p = self.model.transition(s, a)
        r = self.model.reward(s, a, s_p)
        self.agent_state = s_p
        self._mark_done(s_p, a)
        return r, s_p


class Agent:
    """Greedy agent on a solved MDP (``src/mdp.py:1980-2200``)."""

    def __init__(self, model: Model, value_function: Union[ValueFunction, None] = None) -> None:
        self.model = model
        self.value_function = value_function

    def train(self, solver: Union[Solver, None] = None) -> SolverHistory:
        solver = VI_Solver() if solver is None else solver
        self.value_function, hist = solver.solve(self.model)
        return hist

    def get_best_action(self, state: int) -> int:
        assert self.value_function is not None, "No value function, training probably has to be run..."
        best = int(np.argmax(self.value_function.alpha_vector_array[:, state]))
        return int(self.value_function.actions[best])

    def simulate(self, simulator: Union[Simulation, None] = None, max_steps: int = 1000,
                 start_state: Union[int, None] = None, print_progress: bool = True,
                 print_stats: bool = True) -> SimulationHistory:
        simulator = Simulation(self.model) if simulator is None else simulator
        s = simulator.initialize_simulation(start_state=start_state)
        history = SimulationHistory(self.model, s)
        t0 = datetime.now()
        for _ in range(max_steps):
            a = self.get_best_action(s)
            r, s = simulator.run_action(a)
            history.add(action=a, next_state=s, reward=r)
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
                          start_state: int = -1, reward_discount: float = 0.99, print_progress: bool = True,
                          print_stats: bool = True):
        simulator = Simulation(self.model) if simulator is None else simulator
        t0 = datetime.now()
        totals, histories, discounted = RewardSet(), [], []
        for _ in range(n):
            h = self.simulate(simulator, max_steps, start_state, False, False)
            histories.append(h)
            totals.append(sum(h.rewards))
            discounted.append(h.rewards.get_total_discounted_reward(reward_discount))
        if print_stats:
            print(f'All {n} simulations done:')
            print(f'\t- Average runtime (s): {(datetime.now() - t0).total_seconds() / n}')
            print(f'\t- Average step count: {sum(len(h) for h in histories) / n}')
            print(f'\t- Average total rewards: {sum(totals) / n}')
            print(f'\t- Average discounted rewards (ADR): {sum(discounted) / n}')
        return totals, histories
