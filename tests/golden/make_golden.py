"""Generate the golden fixtures by running the REFERENCE itself.

Run in the build container only (needs /root/reference, which never travels):

    MPLBACKEND=Agg python tests/golden/make_golden.py [small] [kat] [c2] [full] [full_r5] [c5] [sim] [models] [solve] [e2e]
                                                      [hsvi] [prune] [limiter]

(`full_r5`: the R = 5 variant at V = B = 1024, the size bench.py's secondary.c4_r5 runs at -- the reference's backup needs
~30 GB and 173 s for it here.)

It imports ``/root/reference/src/pomdp.py`` (NumPy path; CuPy is absent), feeds it
inputs produced by this repo's own deterministic generator
(``pomdp_pbvi_exploration_amd/synth.py``) or the reference's example ``.POMDP``
files, and stores inputs + the reference's outputs as small ``.npz`` / ``.json``
files next to this script.  Fixtures are data only: no reference source text.

The oracle (``oracle/pbvi_oracle.py``) is asserted bit-identical to the reference
here on every case before its per-belief intermediates (best_alpha_ind, which the
reference does not return) are added to a fixture.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
REF = '/root/reference'
EXAMPLES = os.path.join(REF, 'Experiments', 'Example Models')


def _import_reference():
    """Import the reference's ``src`` package without shadowing by this repo's ``src`` shim."""
    for k in [k for k in sys.modules if k == 'src' or k.startswith('src.')]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import src.pomdp as ref_pomdp
            import src.mdp as ref_mdp
    finally:
        sys.path.remove(REF)
    assert ref_pomdp.__file__.startswith(REF), ref_pomdp.__file__
    return ref_pomdp, ref_mdp


ref, ref_mdp = _import_reference()
sys.path.insert(0, REPO)
from oracle import pbvi_oracle as orc                     # noqa: E402
from pomdp_pbvi_exploration_amd import synth              # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def ref_model_from_synth(m: synth.SynthModel):
    """Reference ``Model`` over the synthetic olfactory tables.  R=1: the reference builds
    RTO / ER itself and they must equal synth's bit for bit.  R=5: the per-slot
    probabilities cannot be passed through the constructor, so the three tables the
    backup reads are assigned (they are inputs of the path)."""
    model = quiet(ref.Model, states=m.S, actions=m.A, observations=m.O, reachable_states=m.reachable_states,
                  observation_table=m.observation_table, end_states=[m.goal],
                  start_probabilities=list(m.start_belief))
    if m.R == 1:
        assert np.array_equal(model.reachable_transitional_observation_table, m.rto)
        assert np.array_equal(model.expected_rewards_table, m.expected_rewards)
    else:
        model.reachable_probabilities = m.reachable_probabilities
        model.reachable_transitional_observation_table = m.rto
        model.expected_rewards_table = m.expected_rewards
    return model


def ref_backup(model, alpha, actions, beliefs, gamma, append, prune):
    solver = ref.PBVI_Solver(gamma=gamma)
    vf = ref.ValueFunction(model, alpha, actions)
    bs = ref.BeliefSet(model, beliefs)
    out = solver.backup(model, bs, vf, append=append, belief_dominance_prune=prune)
    return out.alpha_vector_array, np.asarray(out.actions, dtype=np.int64)


def check_oracle(alpha, actions, beliefs, rs, rto, er, gamma, append, prune, ref_rows, ref_acts):
    rows, acts = orc.backup(alpha, actions, beliefs, rs, rto, er, gamma, append=append, belief_dominance_prune=prune)
    assert rows.shape == ref_rows.shape, (rows.shape, ref_rows.shape)
    assert np.array_equal(rows, ref_rows), float(np.max(np.abs(rows - ref_rows)))
    assert np.array_equal(acts, ref_acts)


# --------------------------------------------------------------------------- #
def gen_small():
    """Synthetic olfactory models at S=600 (R=1 and R=5): full inputs + reference outputs."""
    for R in (1, 5):
        m = synth.olfactory_model(H=15, W=40, R=R)
        model = ref_model_from_synth(m)
        alpha, acts = synth.alpha_set(m, 48)
        beliefs = synth.belief_points(m, 64, max_depth=16)
        out = {}
        for tag, append, prune in (('plain', False, False), ('prune', False, True), ('append', True, False)):
            rows, ra = ref_backup(model, alpha, acts, beliefs, m.gamma, append, prune)
            check_oracle(alpha, acts, beliefs, m.reachable_states, m.rto, m.expected_rewards, m.gamma, append, prune, rows, ra)
            out[f'{tag}_alpha'] = rows
            out[f'{tag}_actions'] = ra
        a_new, a_star, v_star = orc.backup_core(alpha, beliefs, m.reachable_states, m.rto, m.expected_rewards, m.gamma)
        keep = orc.belief_dominance_mask(alpha, beliefs, a_new)
        # the tiled oracle (used at |S|~30k) must agree with the untiled one
        t_new, t_star, t_v = orc.backup_core_tiled(alpha, beliefs, m.reachable_states, m.rto, m.expected_rewards, m.gamma, v_tile=16)
        assert np.array_equal(t_star, a_star) and np.array_equal(t_v, v_star) and np.allclose(t_new, a_new, rtol=1e-13, atol=0)
        np.savez_compressed(os.path.join(HERE, f'olfactory_small_R{R}.npz'),
                            H=m.H, W=m.W, R=R, gamma=m.gamma,
                            reachable_states=m.reachable_states.astype(np.int32), rto=m.rto.astype(np.float32),
                            expected_rewards=m.expected_rewards.astype(np.float32),
                            alpha=alpha.astype(np.float32), alpha_actions=acts.astype(np.int8),
                            beliefs=beliefs.astype(np.float32),
                            core_alpha=a_new, core_actions=a_star.astype(np.int8), core_best=v_star.astype(np.int16),
                            core_keep=keep, **out)
        print(f'small R={R}: |V_out| plain={len(out["plain_actions"])} prune={len(out["prune_actions"])} append={len(out["append_actions"])}')


# --------------------------------------------------------------------------- #
def two_state_model(obs_rnd: float):
    """The 2-state model of observation_variation_comparisson.ipynb cells [3]-[6]."""
    T = np.zeros((2, 2, 2))
    for s in range(2):
        for a in range(2):
            for sp in range(2):
                T[s, a, sp] = 0.8 if (s + a) % 2 == sp else round((1.0 - 0.8) / 1, 1)
    Ob = np.zeros((2, 2, 2))
    for sp in range(2):
        for a in range(2):
            for o in range(2):
                Ob[sp, a, o] = obs_rnd if sp == o else (1.0 - obs_rnd) / 1
    Rw = np.zeros((2, 2, 2, 2))
    for sp in range(2):
        Rw[:, :, sp, :] = [0.2, 0.6][sp]
    return T, Ob, Rw


def gen_kat():
    kat = {}
    # KAT-1/2/3: tiger
    model, solver = quiet(ref.load_POMDP_file, os.path.join(EXAMPLES, 'tiger.95.POMDP'))
    vf0 = ref.ValueFunction(model, model.expected_rewards_table.T, model.actions)
    one = solver.backup(model, ref.BeliefSet(model, [ref.Belief(model)]), vf0, belief_dominance_prune=False)
    kat['kat1_alpha'] = one.alpha_vector_array.tolist()
    kat['kat1_actions'] = [int(a) for a in one.actions]
    kat['kat2_belief'] = ref.Belief(model).update(0, 0).values.tolist()
    model.end_actions = [1, 2]
    vf, hist = quiet(solver.solve, model, expansions=8, update_passes=8, print_progress=False)
    kat['kat3_belief_counts'] = [int(c) for c in hist.beliefs_counts]
    kat['kat3_alpha'] = vf.alpha_vector_array.tolist()
    kat['kat3_actions'] = [int(a) for a in vf.actions]
    kat['kat3_alpha_counts'] = [int(c) for c in hist.alpha_vector_counts]
    np.savez_compressed(os.path.join(HERE, 'tiger_tables.npz'),
                        reachable_states=model.reachable_states, reachable_probabilities=model.reachable_probabilities,
                        rto=model.reachable_transitional_observation_table, expected_rewards=model.expected_rewards_table,
                        observation_table=model.observation_table, start=model.start_probabilities, gamma=solver.gamma)

    # KAT-4/5: 2-state model, repeated backup with belief-dominance prune on
    def run(obs_rnd, belief_rows, eps):
        T, Ob, Rw = two_state_model(obs_rnd)
        m = quiet(ref.Model, states=['s0', 's1'], actions=['stay', 'move'], observations=['s0', 's1'], transitions=T,
                  rewards=Rw, observation_table=Ob, rewards_are_probabilistic=True)
        s = ref.PBVI_Solver(gamma=0.99)
        bs = ref.BeliefSet(m, [ref.Belief(m, np.array(r)) for r in belief_rows])
        v = ref.ValueFunction(m, m.expected_rewards_table, m.actions)
        limit = eps * (0.99 / (1 - 0.99))
        old = None
        its = 1000
        for it in range(1000):
            v = s.backup(m, bs, v)
            cur = np.max(np.matmul(bs.belief_array, v.alpha_vector_array.T), axis=1)
            if old is not None and np.max(np.abs(cur - old)) < limit:
                its = it
                break
            old = cur
        return dict(max=float(np.max(v.alpha_vector_array)), min=float(np.min(v.alpha_vector_array)), iters=its,
                    n=len(v), alpha=v.alpha_vector_array.tolist(), actions=[int(a) for a in v.actions])

    kat['kat4'] = {}
    for i in (0, 17, 50):
        kat['kat4'][str(i)] = run(0.7, [[i / 100, 1.0 - (i / 100)], [1.0 - (i / 100), i / 100]], 0.0001)
    all102 = []
    for i in range(51):
        all102 += [[i / 100, 1.0 - (i / 100)], [1.0 - (i / 100), i / 100]]
    kat['kat5'] = {str(acc): run(acc, all102, 0.001) for acc in (0.5, 0.7, 1.0)}
    with open(os.path.join(HERE, 'kat.json'), 'w') as fh:
        json.dump(kat, fh, indent=1)
    print('kat3 belief counts', kat['kat3_belief_counts'], '|V|', len(kat['kat3_actions']))
    print('kat4', {k: (v['max'], v['min'], v['iters']) for k, v in kat['kat4'].items()})
    print('kat5', {k: (v['max'], v['min'], v['iters'], v['n']) for k, v in kat['kat5'].items()})


# --------------------------------------------------------------------------- #
def gen_c2():
    """4x3 grid (BASELINE config 1): seeded FSVI run of the reference; every backup's
    inputs and outputs are stored."""
    path = os.path.join(EXAMPLES, '4x3.95-no_loop_2_grid.POMDP')
    model, _ = quiet(ref.load_POMDP_file, path)
    np.savez_compressed(os.path.join(HERE, 'grid4x3_tables.npz'),
                        reachable_states=model.reachable_states, reachable_probabilities=model.reachable_probabilities,
                        rto=model.reachable_transitional_observation_table, expected_rewards=model.expected_rewards_table,
                        observation_table=model.observation_table, start=model.start_probabilities)
    solver = ref.FSVI_Solver(gamma=0.95, eps=1e-6)
    calls = []
    orig = ref.PBVI_Solver.backup

    def spy(self, mdl, belief_set, value_function, append=False, belief_dominance_prune=True):
        a_in = np.array(value_function.alpha_vector_array)
        act_in = np.array(value_function.actions, dtype=np.int64)
        b_in = np.array(belief_set.belief_array)
        out = orig(self, mdl, belief_set, value_function, append=append, belief_dominance_prune=belief_dominance_prune)
        calls.append(dict(alpha=a_in, actions=act_in, beliefs=b_in, append=append, prune=belief_dominance_prune,
                          out_alpha=np.array(out.alpha_vector_array), out_actions=np.array(out.actions, dtype=np.int64)))
        return out

    ref.PBVI_Solver.backup = spy            # observe the reference's own calls; nothing is altered
    try:
        np.random.seed(0)
        random.seed(0)
        model.end_states = [3, 6]
        vf, hist = quiet(solver.solve, model, expansions=10, max_belief_growth=10, print_progress=False)
    finally:
        ref.PBVI_Solver.backup = orig
    rs, rto, er = model.reachable_states, model.reachable_transitional_observation_table, model.expected_rewards_table
    store = {'n_calls': len(calls), 'gamma': 0.95, 'alpha_counts': np.array(hist.alpha_vector_counts)}
    for i, c in enumerate(calls):
        check_oracle(c['alpha'], c['actions'], c['beliefs'], rs, rto, er, 0.95, c['append'], c['prune'],
                     c['out_alpha'], c['out_actions'])
        a_new, a_star, v_star = orc.backup_core(c['alpha'], c['beliefs'], rs, rto, er, 0.95)
        keep = orc.belief_dominance_mask(c['alpha'], c['beliefs'], a_new)
        # direct-call pattern of the notebooks (prune on, append off) on the same inputs
        p_rows, p_acts = ref_backup(model, c['alpha'], c['actions'], c['beliefs'], 0.95, False, True)
        check_oracle(c['alpha'], c['actions'], c['beliefs'], rs, rto, er, 0.95, False, True, p_rows, p_acts)
        for k, v in dict(alpha=c['alpha'], actions=c['actions'], beliefs=c['beliefs'], append=np.array(c['append']),
                         out_alpha=c['out_alpha'], out_actions=c['out_actions'], core_alpha=a_new, core_actions=a_star,
                         core_best=v_star, core_keep=keep, prune_alpha=p_rows, prune_actions=p_acts).items():
            store[f'c{i}_{k}'] = v
    np.savez_compressed(os.path.join(HERE, 'grid4x3_fsvi.npz'), **store)
    print('c2: calls', len(calls), '|V| trajectory', hist.alpha_vector_counts)


# --------------------------------------------------------------------------- #
def gen_full(cases=((1, 1024, 1024, 'olfactory_full_R1.npz'), (5, 512, 512, 'olfactory_full_R5.npz'))):
    """|S|=30000, V=B=1024 (BASELINE configs 2-3): only OUTPUT summaries are stored;
    both sides regenerate the inputs from synth (checksums pin them)."""
    for R, V, B, fname in cases:
        t0 = time.time()
        m = synth.olfactory_model(R=R)
        alpha, acts = synth.alpha_set(m, V)
        beliefs = synth.belief_points(m, B)
        print(f'full R={R}: inputs in {time.time() - t0:.1f}s, belief density {np.mean(beliefs > 0):.3f}', flush=True)
        model = ref_model_from_synth(m)
        t0 = time.time()
        rows, ra = ref_backup(model, alpha, acts, beliefs, m.gamma, False, False)
        t_ref = time.time() - t0
        print(f'  reference backup {t_ref:.1f}s -> |V_out| {len(ra)}', flush=True)
        a_new, a_star, v_star = orc.backup_core_tiled(alpha, beliefs, m.reachable_states, m.rto, m.expected_rewards, m.gamma)
        d_rows, d_acts = orc.dedup_rows(a_new, a_star)
        assert d_rows.shape == rows.shape and np.array_equal(d_acts, ra)
        assert np.allclose(d_rows, rows, rtol=1e-12, atol=0)
        keep = orc.belief_dominance_mask(alpha, beliefs, a_new)
        idx = synth.splitmix64(99, np.arange(4096, dtype=np.uint64))
        sb = (idx % np.uint64(B)).astype(np.int64)
        ss = ((idx >> np.uint64(20)) % np.uint64(m.S)).astype(np.int64)
        np.savez_compressed(os.path.join(HERE, fname),
                            R=R, V=V, B=B, gamma=m.gamma, ref_seconds=t_ref, n_unique=len(ra),
                            inputs_sha256=synth.checksum(m.reachable_states, m.rto, m.expected_rewards, alpha, beliefs),
                            core_actions=a_star.astype(np.int8), core_best=v_star.astype(np.int16), core_keep=keep,
                            row_sum=a_new.sum(axis=1), b_dot=np.sum(beliefs * a_new, axis=1),
                            sample_b=sb, sample_s=ss, sample_val=a_new[sb, ss],
                            value_max=orc.max_value_per_belief(alpha, beliefs))
        print(f'  stored; keep={int(keep.sum())}/{B}', flush=True)


def gen_full_r5():
    """The stochastic R=5 variant at the size the benchmark's ``secondary.c4_r5`` runs at: |S|=30000, V=B=1024 (the
    reference's ``backup`` needs ~30 GB of temporaries for it; alone in the 62 GB container)."""
    gen_full(cases=((5, 1024, 1024, 'olfactory_full_R5_1024.npz'),))


def gen_c5():
    """BASELINE config 5's workload, |S|=30000, V=1024, B=8192 (the belief set that is sharded over 8 GPUs): the
    reference's own backup, run here in 8 blocks of 1024 beliefs -- row b of every intermediate depends on belief b
    only (src/pomdp.py:1495-1506), and 8192 beliefs at once would need ~115 GB for its temporaries -- with the oracle
    asserted equal to it block by block.  Stored: per-belief output summaries (as olfactory_full_R1.npz) and the size of
    the de-duplicated alpha set over all 8192 beliefs."""
    R, V, B, blk = 1, 1024, 8192, 1024
    m = synth.olfactory_model(R=R)
    alpha, acts = synth.alpha_set(m, V)
    model = ref_model_from_synth(m)
    a_star, v_star, row_sum, b_dot, vmax = [], [], [], [], []
    idx = synth.splitmix64(98, np.arange(4096, dtype=np.uint64))
    sb = (idx % np.uint64(B)).astype(np.int64)
    ss = ((idx >> np.uint64(20)) % np.uint64(m.S)).astype(np.int64)
    sval = np.zeros(4096)
    import hashlib
    h = hashlib.sha256()
    for a in (m.reachable_states, m.rto, m.expected_rewards, alpha):
        h.update(np.ascontiguousarray(a).tobytes())
    keys = {}
    t_ref = 0.0
    for b0 in range(0, B, blk):
        beliefs = synth.belief_points(m, blk, start=b0)
        h.update(np.ascontiguousarray(beliefs).tobytes())
        t0 = time.time()
        rows, ra = ref_backup(model, alpha, acts, beliefs, m.gamma, False, False)
        t_ref += time.time() - t0
        a_new, a_s, v_s = orc.backup_core_tiled(alpha, beliefs, m.reachable_states, m.rto, m.expected_rewards, m.gamma)
        d_rows, d_acts = orc.dedup_rows(a_new, a_s)
        assert d_rows.shape == rows.shape and np.array_equal(d_acts, ra) and np.allclose(d_rows, rows, rtol=1e-12, atol=0)
        for r, a in zip(d_rows, d_acts):
            keys[r.tobytes()] = int(a)                       # ValueFunction's byte dedup over the whole set
        a_star.append(a_s)
        v_star.append(v_s)
        row_sum.append(a_new.sum(axis=1))
        b_dot.append(np.sum(beliefs * a_new, axis=1))
        vmax.append(orc.max_value_per_belief(alpha, beliefs))
        sel = (sb >= b0) & (sb < b0 + blk)
        sval[sel] = a_new[sb[sel] - b0, ss[sel]]
        print(f'  c5 block {b0 // blk}: reference {t_ref:.0f}s so far, |V_out| block {len(ra)}, global {len(keys)}', flush=True)
    np.savez_compressed(os.path.join(HERE, 'olfactory_c5_B8192.npz'),
                        R=R, V=V, B=B, gamma=m.gamma, ref_seconds=t_ref, n_unique=len(keys), inputs_sha256=h.hexdigest(),
                        core_actions=np.concatenate(a_star).astype(np.int8), core_best=np.concatenate(v_star).astype(np.int16),
                        row_sum=np.concatenate(row_sum), b_dot=np.concatenate(b_dot), sample_b=sb, sample_s=ss, sample_val=sval,
                        value_max=np.concatenate(vmax))
    print(f'[golden] olfactory_c5_B8192.npz: reference {t_ref:.0f}s for 8192 beliefs, {len(keys)} distinct rows', flush=True)


# --------------------------------------------------------------------------- #
def _pack_histories(hists):
    """Ragged per-simulation sequences -> flat arrays + lengths."""
    return dict(n_steps=np.array([len(h.actions) for h in hists], dtype=np.int32),
                states=np.concatenate([np.asarray(h.states, dtype=np.int32) for h in hists]),
                actions=np.concatenate([np.asarray(h.actions, dtype=np.int8) for h in hists]),
                observations=np.concatenate([np.asarray(h.observations, dtype=np.int8) for h in hists]),
                rewards=np.concatenate([np.asarray(h.rewards, dtype=np.float64) for h in hists]))


def gen_sim():
    """Policy evaluation: the reference's Agent.run_n_simulations_parallel / simulate, seeded, on the
    synthetic olfactory models at S=600 with a value function the reference's FSVI solver produced, and single
    simulations on tiger.  Stored: the value function, the seeds and every trajectory."""
    import random as pyrandom
    for R in (1, 5):
        m = synth.olfactory_model(H=15, W=40, R=R, f32=False)      # f64 tables: np.random.choice needs exact row sums
        model = ref_model_from_synth(m)
        np.random.seed(0)
        pyrandom.seed(0)
        solver = ref.FSVI_Solver(gamma=m.gamma, eps=1e-6)
        vf, _ = quiet(solver.solve, model, expansions=12, max_belief_growth=20, print_progress=False)
        alpha = np.array(vf.alpha_vector_array)
        acts = np.asarray(vf.actions, dtype=np.int64)
        agent = ref.Agent(model, vf)
        out = {}
        # R > 1: the reference indexes the sampled successor as potentials[chosen][:, 0, 0] (src/pomdp.py:2928), which
        # raises IndexError once fewer than R simulations are alive; short horizons keep its run inside what it can do.
        for tag, n, steps, seed in ((('par', 96, 80, 123), ('par2', 300, 40, 7)) if R == 1 else
                                    (('par', 96, 12, 123), ('par2', 300, 8, 7))):
            np.random.seed(seed)
            pyrandom.seed(seed)
            totals, hists = quiet(agent.run_n_simulations_parallel, n=n, max_steps=steps, print_progress=False, print_stats=False)
            for k, v in _pack_histories(hists).items():
                out[f'{tag}_{k}'] = v
            out[f'{tag}_totals'] = np.asarray(totals, dtype=np.float64)
            out[f'{tag}_cfg'] = np.array([n, steps, seed])
            print(f'sim R={R} {tag}: n={n} reached={int(sum(t > 0 for t in totals))} mean steps={np.mean(out[tag + "_n_steps"]):.1f}')
        np.random.seed(11)
        pyrandom.seed(11)
        _, hists = quiet(agent.run_n_simulations, n=6, max_steps=60, print_progress=False, print_stats=False)
        for k, v in _pack_histories(hists).items():
            out[f'seq_{k}'] = v
        out['seq_cfg'] = np.array([6, 60, 11])
        np.savez_compressed(os.path.join(HERE, f'olfactory_sim_R{R}.npz'), H=m.H, W=m.W, R=R, gamma=m.gamma,
                            alpha=alpha, alpha_actions=acts.astype(np.int8), **out)
        print(f'sim R={R}: |V|={len(acts)}')

    model, solver = quiet(ref.load_POMDP_file, os.path.join(EXAMPLES, 'tiger.95.POMDP'))
    model.end_actions = [1, 2]
    vf, _ = quiet(solver.solve, model, expansions=8, update_passes=8, print_progress=False)
    agent = ref.Agent(model, vf)
    np.random.seed(5)
    pyrandom.seed(5)
    _, hists = quiet(agent.run_n_simulations, n=40, max_steps=30, print_progress=False, print_stats=False)
    np.savez_compressed(os.path.join(HERE, 'tiger_sim.npz'), alpha=np.array(vf.alpha_vector_array),
                        alpha_actions=np.asarray(vf.actions, dtype=np.int8), cfg=np.array([40, 30, 5]),
                        **_pack_histories(hists))
    print(f'tiger sim: mean steps {np.mean([len(h.actions) for h in hists]):.2f}')


# --------------------------------------------------------------------------- #
MODEL_ATTRS = ('transition_table', 'observation_table', 'immediate_reward_table', 'expected_rewards_table',
               'reachable_states', 'reachable_probabilities', 'reachable_transitional_observation_table',
               'start_probabilities')


def gen_models():
    """Every example .POMDP file the reference ships (Experiments/Example Models): the tables its loader builds, as
    sha256 digests + shapes (pomdp_file_tables.json).  Files its loader rejects are listed with the error; files
    whose tables it builds are stored whether or not they are stochastic (the test decides what to compare)."""
    import glob
    import gzip
    import hashlib
    import shutil
    out = {}
    dst = os.path.join(HERE, 'models')
    for f in sorted(glob.glob(os.path.join(EXAMPLES, '*.POMDP')) + glob.glob(os.path.join(EXAMPLES, 'ejs', '*.POMDP'))):
        name = os.path.relpath(f, EXAMPLES).replace(os.sep, '__')
        if os.path.getsize(f) > 200000:                 # cit.POMDP: keep the fixture small
            with open(f, 'rb') as src, open(os.path.join(dst, name + '.gz'), 'wb') as raw, \
                    gzip.GzipFile(filename='', mode='wb', fileobj=raw, compresslevel=9, mtime=0) as z:   # reproducible bytes
                shutil.copyfileobj(src, z)
        else:
            shutil.copyfile(f, os.path.join(dst, name))
        try:
            model, solver = quiet(ref.load_POMDP_file, f)
        except Exception as e:                          # noqa: BLE001 -- recorded, not handled
            out[name] = {'error': f'{type(e).__name__}: {e}'}
            continue
        entry = {'S': model.state_count, 'A': model.action_count, 'O': model.observation_count,
                 'R': model.reachable_state_count, 'gamma': solver.gamma,
                 'stochastic': bool(np.allclose(model.transition_table.sum(axis=2), 1.0) and
                                    np.allclose(model.observation_table.sum(axis=2), 1.0)), 'tables': {}}
        for attr in MODEL_ATTRS:
            a = np.ascontiguousarray(getattr(model, attr))
            entry['tables'][attr] = {'shape': list(a.shape), 'dtype': str(a.dtype),
                                     'sha256': hashlib.sha256(a.tobytes()).hexdigest()}
        out[name] = entry
    with open(os.path.join(HERE, 'pomdp_file_tables.json'), 'w') as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print({k: ('error' if 'error' in v else ('ok' if v['stochastic'] else 'not stochastic')) for k, v in out.items()})


# --------------------------------------------------------------------------- #
SOLVE_MODELS = ('tiger-grid.POMDP', 'hallway.POMDP', 'cheese.95.POMDP', '4x4.95.POMDP', '4x3.95.POMDP', 'cit.POMDP')


def gen_solve():
    """FSVI solves of the reference on example models with many observations / dense transitions: per model the
    belief-count and |V| trajectories and the final alpha set (cfg = expansions, max_belief_growth; seeds 0)."""
    import random as pyrandom
    out = {}
    for name in SOLVE_MODELS:
        model, solver = quiet(ref.load_POMDP_file, os.path.join(EXAMPLES, name))
        np.random.seed(0)
        pyrandom.seed(0)
        t0 = time.time()
        # (the reference's default 'ssea' expansion divides by zero-probability observations on all of these models
        # and asserts; FSVI walks through actually reachable (state, observation) pairs)
        exps, passes = (4, 12) if name == 'cit.POMDP' else (6, 12)
        fsvi = ref.FSVI_Solver(gamma=solver.gamma, eps=1e-6)
        vf, hist = quiet(fsvi.solve, model, expansions=exps, max_belief_growth=passes, print_progress=False)
        key = name.replace('.POMDP', '').replace('.', '_').replace('-', '_')
        out[f'{key}_alpha'] = np.array(vf.alpha_vector_array)
        out[f'{key}_actions'] = np.asarray(vf.actions, dtype=np.int16)
        out[f'{key}_beliefs'] = np.asarray(hist.beliefs_counts, dtype=np.int32)
        out[f'{key}_alphas'] = np.asarray(hist.alpha_vector_counts, dtype=np.int32)
        out[f'{key}_cfg'] = np.array([exps, passes])
        print(f'{name}: S={model.state_count} |V|={len(vf)} beliefs={hist.beliefs_counts} in {time.time() - t0:.1f}s')
    np.savez_compressed(os.path.join(HERE, 'example_solves.npz'), **out)


# --------------------------------------------------------------------------- #
def gen_e2e():
    """The reference's FSVI solve loop at the headline scale (S=30000 synthetic olfactory model, 40 expansions of
    <= 100 beliefs, seeds 0): belief-count and |V| trajectories, per-backup value-function changes, the value of
    the start belief and a digest of the final alpha set.  Takes minutes on CPU."""
    import hashlib
    import random as pyrandom
    m = synth.olfactory_model(R=1, f32=False)
    model = quiet(ref.Model, states=m.S, actions=m.A, observations=m.O, reachable_states=m.reachable_states,
                  observation_table=m.observation_table, end_states=[m.goal], start_probabilities=list(m.start_belief))
    np.random.seed(0)
    pyrandom.seed(0)
    t0 = time.time()
    solver = ref.FSVI_Solver(gamma=m.gamma, eps=1e-6)
    vf, hist = quiet(solver.solve, model, expansions=40, max_belief_growth=100, print_progress=False)
    alpha = np.array(vf.alpha_vector_array)
    b0 = np.asarray(model.start_probabilities)
    print(f'e2e: |V|={len(vf)} |B|={hist.beliefs_counts[-1]} in {time.time() - t0:.0f}s; backup mean {np.mean(hist.backup_times):.2f}s')
    np.savez_compressed(os.path.join(HERE, 'olfactory_e2e_fsvi40.npz'),
                        beliefs=np.asarray(hist.beliefs_counts, dtype=np.int32),
                        alphas=np.asarray(hist.alpha_vector_counts, dtype=np.int32),
                        changes=np.asarray(hist.value_function_changes, dtype=np.float64),
                        actions=np.asarray(vf.actions, dtype=np.int8),
                        value_b0=float(np.max(alpha @ b0)),
                        row_sums=alpha.sum(axis=1), row_b0=alpha @ b0,
                        sha256=hashlib.sha256(np.ascontiguousarray(alpha).tobytes()).hexdigest(),
                        ref_backup_mean_s=float(np.mean(hist.backup_times)))


# --------------------------------------------------------------------------- #
def gen_hsvi():
    """API rows beside the backup that the notebooks call: the reference's HSVI solve (deterministic: greedy descent
    on the bound gap) on tiger and the 4x3 grid and on the S=600 olfactory model, the sawtooth upper bound evaluated at
    probe beliefs, and PBVI_Solver.test_n_simulations seeded on the S=600 models."""
    import random as pyrandom
    out = {}
    cases = [('tiger', os.path.join(EXAMPLES, 'tiger.95.POMDP'), 12, 8), ('grid4x3', os.path.join(EXAMPLES, '4x3.95-no_loop_2_grid.POMDP'), 10, 10)]
    for key, path, exps, growth in cases:
        model, pbvi = quiet(ref.load_POMDP_file, path)
        solver = ref.HSVI_Solver(gamma=pbvi.gamma, eps=1e-6)
        np.random.seed(0)
        pyrandom.seed(0)
        vf, hist = quiet(solver.solve, model, expansions=exps, max_belief_growth=growth, print_progress=False)
        ub = solver._upper_bound
        out[f'{key}_alpha'] = np.array(vf.alpha_vector_array)
        out[f'{key}_actions'] = np.asarray(vf.actions, dtype=np.int16)
        out[f'{key}_beliefs'] = np.asarray(hist.beliefs_counts, dtype=np.int32)
        out[f'{key}_alphas'] = np.asarray(hist.alpha_vector_counts, dtype=np.int32)
        out[f'{key}_changes'] = np.asarray(hist.value_function_changes, dtype=np.float64)
        out[f'{key}_cfg'] = np.array([exps, growth])
        out[f'{key}_ub_points'] = np.array([b.values for b in ub.beliefs])
        out[f'{key}_ub_values'] = np.array(list(ub.belief_value_mapping.values()), dtype=np.float64)
        rng = np.random.RandomState(3)
        probes = rng.random((16, model.state_count))
        probes /= probes.sum(axis=1)[:, None]
        ub.update()
        out[f'{key}_ub_probes'] = probes
        out[f'{key}_ub_probe_values'] = np.array([ub.evaluate(ref.Belief(model, p)) for p in probes])
        print(f'hsvi {key}: |V|={len(vf)} beliefs={hist.beliefs_counts} ub points={len(ub.beliefs)}')
    for R in (1, 5):
        m = synth.olfactory_model(H=15, W=40, R=R, f32=False)
        model = ref_model_from_synth(m)
        solver = ref.HSVI_Solver(gamma=m.gamma, eps=1e-6)
        np.random.seed(0)
        pyrandom.seed(0)
        vf, hist = quiet(solver.solve, model, expansions=8, max_belief_growth=12, print_progress=False)
        out[f'olf{R}_alpha'] = np.array(vf.alpha_vector_array)
        out[f'olf{R}_actions'] = np.asarray(vf.actions, dtype=np.int16)
        out[f'olf{R}_beliefs'] = np.asarray(hist.beliefs_counts, dtype=np.int32)
        out[f'olf{R}_alphas'] = np.asarray(hist.alpha_vector_counts, dtype=np.int32)
        out[f'olf{R}_cfg'] = np.array([8, 12])
        print(f'hsvi olfactory R={R}: |V|={len(vf)} beliefs={hist.beliefs_counts}')
        # test_n_simulations of the reference, seeded (R=5: short horizon, the reference's successor indexing raises
        # once fewer than R simulations remain -- it never filters here, so any horizon works; keep it short anyway)
        np.random.seed(21)
        pyrandom.seed(21)
        n, horizon = (64, 50) if R == 1 else (64, 12)
        starts, done_at, rewards, disc = quiet(ref.PBVI_Solver(gamma=m.gamma).test_n_simulations, model, vf, n=n, horizon=horizon)
        out[f'olf{R}_tns_cfg'] = np.array([n, horizon, 21])
        out[f'olf{R}_tns_starts'] = np.asarray(starts, dtype=np.int32)
        out[f'olf{R}_tns_done_at'] = np.asarray(done_at, dtype=np.int32)
        out[f'olf{R}_tns_rewards'] = np.asarray(rewards, dtype=np.float64)
        out[f'olf{R}_tns_discounted'] = np.asarray(disc, dtype=np.float64)
        print(f'test_n_simulations R={R}: steps run {len(rewards)}, done {int(np.sum(np.asarray(done_at) >= 0))}/{n}')
        out[f'olf{R}_coords'] = np.array(model.get_coords([0, 41, 599]))
    np.savez_compressed(os.path.join(HERE, 'hsvi_and_rollouts.npz'), **out)


def gen_prune():
    """``ValueFunction.prune(level=2)`` of the reference (src/mdp.py:857-866) on a seeded alpha set at S=600 that holds
    undominated rows, rows dominated by one other row, rows dominating others, a pair that differs only in the sign of
    a zero (different bytes -- both survive the constructor's dedup -- but each is >= the other, so BOTH go) and rows
    equal to another one except in a single state.  Stored: the input rows / actions and what the reference kept."""
    m = synth.olfactory_model(H=15, W=40, R=1)
    base, acts = synth.alpha_set(m, 24)
    u = synth.uniform01(99, np.arange(base.size, dtype=np.uint64)).reshape(base.shape)
    rows = [base]
    rows.append(base[:6] - 0.01 * u[:6])                       # strictly below one other row everywhere -> dominated
    rows.append(base[6:10] + 0.02 * u[6:10])                   # strictly above -> their originals are dominated
    bump = base[10:14].copy()
    bump[np.arange(4), [3, 77, 300, 599]] += 0.5               # above the original in one state, equal elsewhere
    rows.append(bump)
    dip = base[14:16].copy()
    dip[np.arange(2), [5, 410]] -= 0.25                        # below in one state, equal elsewhere -> dominated
    rows.append(dip)
    zpair = np.round(base[16:17], 1)
    zpair[0, 7] = 0.0
    znegv = zpair.copy()
    znegv[0, 7] = -0.0                                         # same values, different bytes
    rows += [zpair, znegv]
    alpha = np.concatenate(rows).astype(np.float32).astype(np.float64)
    actions = (synth.splitmix64(3, np.arange(alpha.shape[0], dtype=np.uint64)) % np.uint64(m.A)).astype(np.int64)
    model = ref_model_from_synth(m)
    vf = quiet(ref.ValueFunction, model, alpha, actions)
    assert len(vf) == alpha.shape[0], 'the constructor must not have removed anything (all rows differ in bytes)'
    quiet(vf.prune, 2)
    kept_rows, kept_actions = np.asarray(vf.alpha_vector_array), np.asarray(vf.actions)
    kept = np.array([int(np.flatnonzero([r.tobytes() == k.tobytes() for r in alpha])[0]) for k in kept_rows])
    assert np.array_equal(alpha[kept], kept_rows) and np.array_equal(actions[kept], kept_actions)
    mask = orc.prune_dominated_mask(alpha)
    assert np.array_equal(np.flatnonzero(mask), kept), 'oracle prune differs from the reference'
    n = alpha.shape[0]
    assert not mask[n - 1] and not mask[n - 2], 'the +0 / -0 pair dominates each other'
    np.savez_compressed(os.path.join(HERE, 'prune_level2.npz'), alpha=alpha, actions=actions, kept=kept)
    print(f'[golden] prune_level2.npz: {n} rows -> reference keeps {len(kept)}')


def gen_limiter():
    """The |V| limiter of the reference's solve loop (src/pomdp.py:2347-2365): seeded FSVI solve of the S=600 olfactory
    model with ``limit_value_function_size`` set, so most backups are followed by the usefulness scan
    (matmul + argmax + unique) and the weighted random deletion.  Stored: belief-count and |V| trajectories, per-backup
    changes, the final alpha set.  The deletion draws from np.random, so the trajectory only reproduces if every
    ``useful`` set along the way is the reference's."""
    import random as pyrandom
    m = synth.olfactory_model(H=15, W=40, R=1, f32=False)
    cfg = dict(expansions=14, max_belief_growth=8, limit_value_function_size=24)
    model = quiet(ref.Model, states=m.S, actions=m.A, observations=m.O, reachable_states=m.reachable_states,
                  observation_table=m.observation_table, end_states=[m.goal], start_probabilities=list(m.start_belief))
    np.random.seed(0)
    pyrandom.seed(0)
    solver = ref.FSVI_Solver(gamma=m.gamma, eps=1e-6)
    vf, hist = quiet(solver.solve, model, cfg['expansions'], max_belief_growth=cfg['max_belief_growth'],
                     limit_value_function_size=cfg['limit_value_function_size'], print_progress=False)
    counts = np.asarray(hist.alpha_vector_counts)
    assert np.any(np.diff(counts) < 0), 'the limiter never fired: pick a tighter limit'
    np.savez_compressed(os.path.join(HERE, 'limiter_fsvi.npz'), cfg=json.dumps(cfg),
                        alpha_counts=counts, belief_counts=np.asarray(hist.beliefs_counts),
                        changes=np.asarray(hist.value_function_changes, dtype=np.float64),
                        final_alpha=np.asarray(vf.alpha_vector_array), final_actions=np.asarray(vf.actions))
    print(f'[golden] limiter_fsvi.npz: |V| trajectory {counts.tolist()}')


if __name__ == '__main__':
    which = sys.argv[1:] or ['small', 'kat', 'c2']
    for w in which:
        {'small': gen_small, 'kat': gen_kat, 'c2': gen_c2, 'full': gen_full, 'full_r5': gen_full_r5, 'sim': gen_sim, 'models': gen_models, 'solve': gen_solve, 'e2e': gen_e2e, 'hsvi': gen_hsvi, 'prune': gen_prune, 'limiter': gen_limiter, 'c5': gen_c5}[w]()
