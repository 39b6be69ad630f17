"""Deterministic synthetic olfactory-navigation workload.

The reference's large models (``Experiments/Olfactory Navigation/
Olfactory_Alternation_Paper_Wrap.ipynb`` cells building a 61x361 wrap-around
grid, :254-328) need ``cv2`` and plume data files that are not available, so the
benchmark and the full-size parity tests use a closed-form plume on a 75x400
wrap-around grid built by the same index rules:

* states ``s = y*W + x``; actions 0..5 = N, E, S, W, stay+sniff-ground,
  stay+sniff-air, moves wrap around the grid (``...Paper_Wrap.ipynb:299-304``);
* observations 0 nothing / 1 something / 2 goal; ``O[s,a,1] = p(s)`` with the
  ground plume for a<5 and the air plume for a=5, the goal state always emits
  observation 2 (``...Paper_Wrap.ipynb:254-266``);
* reward 1 on landing in the goal (``...Paper_Wrap.ipynb:316-317``).

Everything random comes from a counter-based splitmix64 hash so the fixture
generator (build container) and the GPU box regenerate identical bits without
depending on NumPy's random streams.  Tables use only +,*,/ and one ``exp`` and
are rounded to float32-representable values when ``f32=True`` so that an fp64
oracle and the fp32 engine start from the same numbers.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, idx) -> np.ndarray:
    """Counter-based hash: element ``idx`` of stream ``seed`` as uint64."""
    with np.errstate(over='ignore'):
        x = np.uint64(seed) + (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * _GOLD
        z = (x ^ (x >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, idx) -> np.ndarray:
    """U[0,1) doubles with 53 random bits."""
    return (splitmix64(seed, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@dataclass
class SynthModel:
    H: int
    W: int
    S: int
    A: int
    O: int
    R: int
    goal: int
    gamma: float
    reachable_states: np.ndarray          # [S,A,R] int64
    reachable_probabilities: np.ndarray   # [S,A,R] f64
    observation_table: np.ndarray         # [S,A,O] f64
    rto: np.ndarray                       # [S,A,O,R] f64 (reachable_transitional_observation_table)
    expected_rewards: np.ndarray          # [S,A] f64
    start_belief: np.ndarray              # [S] f64


def _round32(x: np.ndarray, f32: bool) -> np.ndarray:
    return x.astype(np.float32).astype(np.float64) if f32 else x


def olfactory_model(H: int = 75, W: int = 400, R: int = 1, gamma: float = 0.99, f32: bool = True) -> SynthModel:
    """Build the closed-form olfactory model (R=1 faithful, R=5 stochastic moves)."""
    assert R in (1, 5)
    S, A, O = H * W, 6, 3
    gy, gx = H // 2, min(60, W // 4)
    goal = gy * W + gx
    y, x = np.divmod(np.arange(S, dtype=np.int64), W)

    def nb(dy, dx):
        return ((y + dy) % H) * W + ((x + dx) % W)

    north, east, south, west, stay = nb(-1, 0), nb(0, 1), nb(1, 0), nb(0, -1), np.arange(S, dtype=np.int64)
    moves = [north, east, south, west, stay]
    intended = [0, 1, 2, 3, 4, 4]                     # index into moves per action
    rs = np.empty((S, A, R), dtype=np.int64)
    rp = np.empty((S, A, R), dtype=np.float64)
    if R == 5:
        assert H >= 3 and W >= 3, "R=5 needs five distinct successors"
    for a in range(A):
        rs[:, a, 0] = moves[intended[a]]
        if R == 1:
            rp[:, a, 0] = 1.0
        else:
            # intended successor 0.8; the four other members of {N,E,S,W,stay} 0.05 each
            rp[:, a, 0] = 0.8
            others = [k for k in range(5) if k != intended[a]]
            for r, k in enumerate(others, start=1):
                rs[:, a, r] = moves[k]
                rp[:, a, r] = 0.05
    rp = _round32(rp, f32)

    # closed-form plume, downwind (+x) of the source at the goal column
    dxp = np.maximum(x - gx, 0).astype(np.float64)
    sig = 3.0 + 0.03 * dxp
    lat = np.exp(-((y - gy).astype(np.float64) ** 2) / (2.0 * sig * sig))

    def plume(L):
        p = 0.7 * lat * np.exp(-dxp / L) * (x >= gx)
        return _round32(p, f32)

    p_ground, p_air = plume(120.0), plume(200.0)
    obs = np.zeros((S, A, O), dtype=np.float64)
    for a in range(A):
        p = p_air if a == 5 else p_ground
        obs[:, a, 1] = p
        obs[:, a, 0] = _round32(1.0 - p, f32)
    obs[goal, :, :] = 0.0
    obs[goal, :, 2] = 1.0

    reach_obs = obs[rs[:, :, None, :], np.arange(A)[None, :, None, None], np.arange(O)[None, None, :, None]]
    rto = _round32(rp[:, :, None, :] * reach_obs, f32)                 # src/pomdp.py:201-202
    er = _round32(np.sum(rp * (rs == goal), axis=2), f32)              # reward 1 on landing in goal

    start = np.zeros((H, W))
    y0, y1 = (H * 20) // 75, (H * 55) // 75
    x0, x1 = gx, max(gx + 1, (W * 360) // 400)
    start[y0:y1, x0:x1] = 1.0
    start = (start / start.sum()).reshape(S)
    return SynthModel(H, W, S, A, O, R, goal, gamma, rs, rp, obs, rto, er, start)


def bayes_update(belief: np.ndarray, a: int, o: int, m: SynthModel) -> np.ndarray:
    """Un-normalised Bayes step (same arithmetic as the reference's
    ``Belief.update``, ``src/pomdp.py:405-408``)."""
    w = m.rto[:, a, o, :] * belief[:, None]
    return np.bincount(m.reachable_states[:, a, :].ravel(), weights=w.ravel(), minlength=m.S)


def belief_points(m: SynthModel, B: int, seed: int = 1, max_depth: int = 64, f32: bool = True,
                  start: int = 0) -> np.ndarray:
    """B beliefs: the start belief pushed through k in [1,max_depth] random
    (action, observation) Bayes updates, o ~ P(o|b,a) (perseus-style walk,
    ``src/pomdp.py:2041-2054``).  Consecutive rows share a trajectory: row i is
    step (i mod max_depth)+1 of walk i // max_depth, so generation costs B
    updates.  ``start`` (a multiple of max_depth) yields rows start..start+B-1 of the same
    global sequence, so each rank of a sharded run generates only its own block."""
    assert start % max_depth == 0
    out = np.empty((B, m.S), dtype=np.float64)
    b = m.start_belief
    for i in range(start, start + B):
        if i % max_depth == 0:
            b = m.start_belief
        a = int(splitmix64(seed, 2 * i) % np.uint64(m.A))
        u = float(uniform01(seed, 2 * i + 1))
        cand = [bayes_update(b, a, o, m) for o in range(m.O)]
        mass = np.array([c.sum() for c in cand])
        cdf = np.cumsum(mass) / mass.sum()
        o = int(np.searchsorted(cdf, u, side='right'))
        o = min(o, m.O - 1)
        while mass[o] == 0.0:                       # exact-zero guard only (no rounding sensitivity)
            o = (o + 1) % m.O
        b = cand[o] / mass[o]
        out[i - start] = _round32(b, f32)
    return out


def dense_belief_points(S: int, B: int, seed: int = 2, f32: bool = True) -> np.ndarray:
    """Uniform-random normalised rows (``expand_ra``, ``src/pomdp.py:1545-1546``)."""
    idx = np.arange(B * S, dtype=np.uint64).reshape(B, S)
    b = uniform01(seed, idx)
    b /= b.sum(axis=1, keepdims=True)
    return _round32(b, f32)


def alpha_set(m: SynthModel, V: int, seed: int = 7, f32: bool = True):
    """V alpha-vectors: row v = V_mdp * u_v + eps_v with V_mdp[s] = gamma^d(s,goal)
    (wrap-around Manhattan distance), u_v ~ U(0.5,1), eps ~ U(0,0.05); actions ~ U{0..A-1}."""
    y, x = np.divmod(np.arange(m.S, dtype=np.int64), m.W)
    gy, gx = divmod(m.goal, m.W)
    dy = np.abs(y - gy)
    dx = np.abs(x - gx)
    d = np.minimum(dy, m.H - dy) + np.minimum(dx, m.W - dx)
    powtab = np.empty(int(d.max()) + 1)
    powtab[0] = 1.0
    for i in range(1, powtab.size):
        powtab[i] = powtab[i - 1] * m.gamma        # plain multiplies: bit-reproducible
    vmdp = powtab[d]
    u = 0.5 + 0.5 * uniform01(seed, np.arange(V, dtype=np.uint64))
    idx = np.arange(V * m.S, dtype=np.uint64).reshape(V, m.S) + np.uint64(1 << 40)
    eps = 0.05 * uniform01(seed, idx)
    alpha = _round32(vmdp[None, :] * u[:, None] + eps, f32)
    actions = (splitmix64(seed, np.arange(V, dtype=np.uint64) + np.uint64(1 << 50)) % np.uint64(m.A)).astype(np.int64)
    return alpha, actions


def sea_robin_like(H: int = 165, W: int = 375, V: int = 1386, B: int = 100, seed: int = 61875):
    """Tables of the SHAPE of the reference's largest other model family (``Experiments/Sea Robins/Sea_Robin_Real.ipynb``:
    S = 61875 = 165 x 375, A = 16, O = 2, one reachable state per (s, a); its CuPy run went out of memory allocating
    Gamma[A,O,V,S] in fp64 at |V| = 1386, ``:913``): 16 grid moves with wrap-around, a random two-way observation model with
    impossible observations, a sparse reward, V alpha-vectors and B sparse beliefs.  Synthetic -- the notebook's data files
    are not shipped.  Returns ``(S, A, O, rs, rto, er, alpha, beliefs)`` (fp32-representable fp64 arrays)."""
    rng = np.random.default_rng(seed)
    A, O = 16, 2
    S = H * W
    y, x = np.divmod(np.arange(S), W)
    moves = [(-1, 0), (0, 1), (1, 0), (0, -1), (-2, 0), (0, 2), (2, 0), (0, -2), (-1, 1), (1, 1), (1, -1), (-1, -1), (0, 0),
             (0, 5), (5, 0), (0, -5)]
    rs = np.stack([((y + dy) % H) * W + (x + dx) % W for dy, dx in moves], axis=1)[:, :, None].astype(np.int64)
    p = 0.05 + 0.9 * rng.random((S, A))
    p[rng.random((S, A)) < 0.2] = 0.0                                  # states where one observation is impossible
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    rto = r32(np.stack([p, 1.0 - p], axis=2)[:, :, :, None])
    er = (rng.random((S, A)) < 0.01).astype(np.float64)
    alpha = r32(rng.random((V, S)) * rng.random((V, 1)) * 10.0)
    b = rng.random((B, S)) * (rng.random((B, S)) < 0.1)
    b[:, 17] += 1e-3
    beliefs = r32(b / b.sum(axis=1, keepdims=True))
    return S, A, O, rs, rto, er, alpha, beliefs


def checksum(*arrays) -> str:
    """sha256 over the raw bytes of the given arrays (fixture input pin)."""
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
