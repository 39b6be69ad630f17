"""Headline benchmark: alpha-vector backups / second at |S|=30000, |B|=1024 per GPU.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1)

A step is ONE point-based backup of SURVEY 8d's metric, of a belief block the engine has NOT seen as a block before:
``pbvi_beliefs_select`` of the next block of B rows of the device belief store (rotating id ranges: the rows are in
HBM, no PCIe, but everything the engine derives from a belief block -- sort order, zero-tile map, per-belief tile lists,
dead-triple flags -- is built inside the step, as in a solve, which backs each new block up exactly once), then
``pbvi_backup_run`` (Gamma projection, score GEMM, argmax + fp64 tie refinement, action selection, dedup, alpha'
assembly) against the resident alpha set, THEN its results -- the U distinct alpha' rows, the per-belief index into them
and the actions -- copied into page-locked host buffers (``pbvi_backup_fetch_compact``), synchronised.  That is what the
reference's own ``backup_times`` contain: ``ValueFunction.__init__``'s ``tobytes()`` forces the device-to-host copy
(src/mdp.py:667-669).  ``value`` = beliefs / MEDIAN step time (median of the timed calls).  Beside it:
``value_reused_block`` (round 2's ``value``: the same block backed up again and again, its indexes kept), the mean over
the whole timed region and the device-resident figure (results left in HBM).

With N>1 the beliefs are sharded and each step is the product path of ``dist.sharded_engine_step``: local backup, ONE
all-gather of integers (per-belief index / action / keep + the keys of the distinct rows), global dedup, and every
replica appends the globally distinct rows to its alpha store.  ``--scaling weak`` (default) keeps ``--beliefs`` per
GPU; ``--scaling strong --beliefs-total 8192`` is BASELINE config 5 literally.

Workload: the synthetic olfactory model of SURVEY.md 8d, reachable-sparse R=1 (BASELINE config "reachable-sparse CSR
SpMM backup", as every large model of the reference).  The other BASELINE configurations that fit one GPU run in the
same process and are reported under ``secondary``, each with its own roofline: ``c3_dense`` (dense projection as
|A||O| MFMA GEMMs over densified T.O, every tile multiplied), ``c4_f64`` (the reference's precision), ``c4_r5``
(stochastic moves, 5 reachable states).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F64_MFMA_TFLOPS = 78.6       # MI355X_MICROARCH.md: v_mfma_f64_16x16x4_f64, dense
PEAK_HBM_GBS = 8000.0
STAGES = ('ms_total', 'ms_project', 'ms_score', 'ms_argmax', 'ms_refine', 'ms_action', 'ms_assemble')


def cpu_baseline(m, alpha, beliefs, sample: int):
    """The oracle (NumPy restatement of src/pomdp.py:1485-1506, fp64) timed on this host's cores
    on a bounded sample of the same workload: the first ``sample`` beliefs, the full alpha set."""
    from oracle import pbvi_oracle as orc            # cpu_baseline leg only
    b = beliefs[:sample]
    t0 = time.perf_counter()
    orc.backup_core(alpha, b, m.reachable_states, m.rto, m.expected_rewards, m.gamma)
    dt = time.perf_counter() - t0
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    blas_threads = None
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max((p.get('num_threads', 0) for p in threadpool_info() if p.get('user_api') == 'blas'), default=None)
    except Exception:
        pass
    return {'value': sample / dt, 'unit': 'backups/s', 'cores': blas_threads or cores, 'kind': 'port',
            'sample': f'first {sample} of the {beliefs.shape[0]} beliefs x all {alpha.shape[0]} alpha-vectors, '
                      f'untiled NumPy fp64 statements (oracle/pbvi_oracle.py::backup_core), 1 call, {dt:.1f} s; '
                      f'{cores} schedulable cores, OpenBLAS threads {blas_threads}'}


_PMC = None


def load_pmc():
    """profiles/r03_pmc_traffic.json: {configuration: {kernel, traffic_bytes, kernels: {name: {...}}}} or {}."""
    global _PMC
    if _PMC is None:
        try:
            with open(os.path.join(REPO, 'profiles', 'r03_pmc_traffic.json')) as fh:
                _PMC = json.load(fh)
        except (OSError, ValueError):
            _PMC = {}
    return _PMC


def attach_traffic(roofline, tag):
    """`traffic` of a roofline entry: PMC bytes per launch of its kernel, measured on this configuration by rocprofv3."""
    pmc = load_pmc().get(tag)
    if pmc and pmc.get('traffic_bytes') and roofline is not None and pmc['kernel'].split('<')[0] in roofline['kernel']:
        roofline['traffic'] = pmc['traffic_bytes']
        roofline['traffic_source'] = f'profiles/r03_pmc_traffic.json[{tag}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)'


def stage_rooflines(stage_ms, tag):
    """The HBM-bound stages beside the GEMM: PMC bytes per launch of the stage's main kernel (rocprofv3 passes on this
    configuration, profiles/r03_pmc_traffic.json) over the stage's live time from this run.  The stage time also holds its
    small helper kernels and, under pbvi_backup_run_fetch, what runs beside it (the rows' PCIe copy beside the refinement),
    so these fractions are lower bounds of the kernels' own."""
    pmc = load_pmc().get(tag)
    if not pmc or not stage_ms:
        return None
    sec = {}
    for stage, prefix in (('ms_project', 'k_project<'), ('ms_argmax', 'k_argmax<'), ('ms_refine', 'k_refine<')):
        kern = next((k for k in pmc['kernels'] if k.startswith(prefix)), None)
        ms = stage_ms.get(stage, 0.0)
        if kern is None or ms <= 0.0:
            continue
        tb = pmc['kernels'][kern]['traffic_bytes']
        if tb < 50e6:                                          # (a residual launch, e.g. the fused engines' one projected tile)
            continue
        gbs = tb / (ms * 1e-3) / 1e9
        sec[kern] = {'stage': stage, 'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS,
                     'traffic': tb, 'ms': ms}
    return sec or None


class HostResults:
    """Page-locked destination of one step's results (pbvi_host_alloc): U rows (room for B), index, actions."""

    def __init__(self, eng, B: int):
        from pomdp_pbvi_exploration_amd.engine import PinnedBuffer
        item = 4 if eng.dtype == 'f32' else 8
        self.buf = PinnedBuffer(B * eng.S * item + 3 * B * 4 + 8192)
        self.rows = self.buf.carve((B, eng.S), eng.np_dtype)
        self.index = self.buf.carve((B,), np.int32)
        self.actions = self.buf.carve((B,), np.int32)
        self.slot = self.buf.carve((B,), np.int32)           # pbvi_backup_run_fetch: row of distinct key u is rows[slot[u]]

    def close(self):
        self.rows = self.index = self.actions = self.slot = None
        self.buf.close()


def gemm_roofline(stats, dtype: str, mode: str):
    """Roofline of the dominant kernel from the engine's per-call stats (HIP events on the engine's stream)."""
    if mode == 'dense':     # the projection GEMMs dominate
        ms = float(np.mean([s['ms_project_gemm'] for s in stats]))
        pf, pfe = stats[0]['project_flops'], stats[0]['project_flops_executed']
        ach = pfe / (ms * 1e-3) / 1e12
        return {'bound': 'mfma', 'kernel': 'k_gemm_nt_f32_mfma (dense projection alpha x D_ao^T, batched over (a,o))',
                'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS,
                'traffic': None, 'flops_per_launch': pfe, 'ms_per_launch': ms, 'dense_flops_per_launch': pf}
    ms = float(np.mean([s['ms_score'] for s in stats]))
    # (steps may back different belief blocks up: the executed tile-steps are averaged like the launch times)
    flops, dense = float(np.mean([s['score_flops_executed'] for s in stats])), stats[0]['score_flops']
    f32_gemm = dtype == 'f32' or bool(stats[0].get('screened'))      # an fp64 engine's scores come from its fp32 screen
    peak = PEAK_F32_MFMA_TFLOPS if f32_gemm else PEAK_F64_MFMA_TFLOPS
    ach = flops / (ms * 1e-3) / 1e12
    fused = bool(stats[0].get('fused_projection'))
    return {'bound': 'mfma', 'kernel': (('k_gemm_nt_f32_streamk_fused' if fused else 'k_gemm_nt_f32_streamk') if f32_gemm
                                        else 'k_gemm_nt_f64_mfma') +
                                       ' (belief x Gamma score GEMM, non-zero tiles' +
                                       ('; Gamma tiles generated in the operand staging instead of read from HBM: the kernel '
                                        'also does the work of the projection kernel it replaces' if fused else '') +
                                       (', fp32 screen of an fp64 engine)' if f32_gemm and dtype != 'f32' else ')'),
            'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
            'flops_per_launch': flops, 'ms_per_launch': ms, 'dense_flops_per_launch': dense,
            'dense_equivalent_tflops': dense / (ms * 1e-3) / 1e12,
            'tiles_run_over_dense': float(np.mean([s['score_tiles_run'] for s in stats])) / max(1, stats[0]['score_tiles_dense'])}


class stdout_to_stderr:
    """RCCL prints a version banner on stdout when its first communicator comes up; the driver expects ONE JSON line
    there.  File descriptor 1 points at stderr while the process group and the warm-up collectives are set up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def timed_steps(step, steps: int, warmup: int, fence):
    """``warmup`` untimed steps, then ``steps`` timed ones between two fences.  Returns (per-step seconds, whole-region
    seconds, list of what step() returned)."""
    for _ in range(warmup):
        step()
    fence()
    per, outs = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        t = time.perf_counter()
        outs.append(step())
        per.append(time.perf_counter() - t)
    fence()
    return np.asarray(per), time.perf_counter() - t0, outs


def measure_config(name, m, alpha, beliefs, dtype, mode, steps, warmup, fence, true_dense=False, screen=None, fused=None,
                   formulation=None, pmc_tag=None):
    """One single-GPU configuration measured like round 2's headline (run + results to pinned host, median; the SAME
    belief block every step -- ``reused_block`` in the entry -- the fresh-block protocol is the headline's)."""
    from pomdp_pbvi_exploration_amd.engine import Engine, debug_gemm_dense
    prev = debug_gemm_dense(True) if true_dense else None
    try:
        eng = Engine(m.S, m.A, m.O, m.R, m.reachable_states, m.rto, m.expected_rewards, dtype=dtype, mode=mode)
        if formulation is not None:
            eng.set_formulation(formulation)
        if screen is not None:
            eng.set_f64_screen(screen)
        if fused is not None:
            eng.set_fused_projection(fused)
        eng.set_alpha(alpha)
        eng.set_beliefs(beliefs)
        B = beliefs.shape[0]
        host = HostResults(eng, B)

        def step():
            st, _, _ = eng.run_fetch_into(m.gamma, host.rows, host.slot, host.index, host.actions)
            return st

        per, _, stats = timed_steps(step, steps, warmup, fence)
        med = float(np.median(per))
        out = {'workload': name, 'dtype': dtype, 'value': B / med, 'unit': 'backups/s', 'ms_per_step': med * 1e3,
               'steps': steps, 'warmup': warmup, 'reused_block': True, 'formulation': {1: 'alpha-side', 2: 'belief-side'}.get(int(stats[-1].get('formulation', 0)), '?'),
               'S': m.S, 'A': m.A, 'O': m.O, 'R': m.R, 'V': int(alpha.shape[0]), 'B': B,
               'roofline': gemm_roofline(stats, dtype, mode),
               'stage_ms': {k: float(np.mean([s[k] for s in stats])) for k in STAGES},
               'unique_rows': int(stats[-1]['n_unique'])}
        if pmc_tag:
            attach_traffic(out['roofline'], pmc_tag)
            sr = stage_rooflines(out['stage_ms'], pmc_tag)
            if sr:
                out['stage_roofline'] = sr
        host.close()
        eng.close()
        return out
    finally:
        if prev is not None:
            debug_gemm_dense(prev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--beliefs', type=int, default=1024, help='beliefs per GPU (weak scaling)')
    ap.add_argument('--scaling', type=str, default='weak', choices=['weak', 'strong'])
    ap.add_argument('--beliefs-total', type=int, default=8192, help='beliefs over all GPUs (--scaling strong: BASELINE config 5)')
    ap.add_argument('--alphas', type=int, default=1024)
    ap.add_argument('--reach', type=int, default=1, choices=[1, 5], help='reachable states per (s,a)')
    ap.add_argument('--cpu-sample', type=int, default=1024, help='beliefs in the CPU baseline sample (0 = skip)')
    ap.add_argument('--grid', type=str, default='75x400')
    ap.add_argument('--dtype', type=str, default='f32', choices=['f32', 'f64'], help='engine arithmetic type')
    ap.add_argument('--formulation', type=str, default='auto', choices=['auto', 'alpha', 'belief'],
                    help='operand projected through the model (pbvi_set_formulation)')
    ap.add_argument('--mode', type=str, default='sparse', choices=['sparse', 'dense'],
                    help="projection: reachable-sparse ELL SpMM (BASELINE config 3, the reference's path) or dense "
                         "|A||O| MFMA GEMMs over densified T.O (config 2; 65 GB of matrices at S=30000)")
    ap.add_argument('--two-calls', action='store_true', help='time pbvi_backup_run + pbvi_backup_fetch_compact (round 2\'s step) instead '
                                                            'of pbvi_backup_run_fetch')
    ap.add_argument('--blocks', type=int, default=3, help='belief blocks in the device store the timed steps rotate over')
    ap.add_argument('--secondary', type=str, default='auto', choices=['auto', 'none'],
                    help="'auto': on one GPU with the default workload also measure c3_dense, c4_f64, c4_r5")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N')
    # PBVI_DIST_BACKEND=gloo: a rehearsal of the sharded step with several ranks on ONE GPU (RCCL refuses two ranks on one
    # device): every rank's engine lives on device 0 and the exchange travels through host tensors -- the per-rank backup
    # times then share the card and mean nothing, the exchange / merge / append split is the product path's.
    backend = os.environ.get('PBVI_DIST_BACKEND', 'nccl')
    if backend == 'gloo' and torch.cuda.device_count() < world:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or os.environ.get('PBVI_FORCE_DIST') == '1'   # the latter: rehearse the RCCL path on one GPU
    dist = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        with stdout_to_stderr():
            if backend == 'gloo':
                dist.init_process_group('gloo')
            else:
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            dist.barrier()                                          # brings the communicator (and its banner) up now

    from pomdp_pbvi_exploration_amd import synth
    from pomdp_pbvi_exploration_amd.engine import Engine
    from pomdp_pbvi_exploration_amd.dist import EngineShard, gather_unique, shard_bounds, sharded_engine_step

    H, W = (int(x) for x in args.grid.split('x'))
    m = synth.olfactory_model(H=H, W=W, R=args.reach)
    alpha, _ = synth.alpha_set(m, args.alphas)
    if args.scaling == 'strong':
        n_total = args.beliefs_total
        lo, hi, per = shard_bounds(n_total, world, rank)
        if per % 64:
            raise SystemExit('--beliefs-total / --gpus must be a multiple of 64 (the synthetic walks restart every 64 beliefs)')
        B = hi - lo
        beliefs = synth.belief_points(m, B, start=lo)
    else:
        B = args.beliefs
        n_total = B * world
        beliefs = synth.belief_points(m, B, start=rank * B)        # this rank's block of the global set

    exchange_rows_env = os.environ.get('PBVI_EXCHANGE') == 'rows'   # A-B only: move alpha' rows instead of keys
    eng = Engine(m.S, m.A, m.O, m.R, m.reachable_states, m.rto, m.expected_rewards, dtype=args.dtype, device=local_rank,
                 mode=args.mode)
    eng.set_formulation(args.formulation)
    eng.set_alpha(alpha)
    # the belief store: `blocks` blocks of B rows (block 0 = `beliefs`); a timed step selects the next one by id
    rotate = args.blocks > 1 and not exchange_rows_env
    block_ids = []
    if rotate:
        for k in range(args.blocks):
            rows = beliefs if k == 0 else synth.belief_points(m, B, start=(rank * args.blocks + k) * B if args.scaling == 'weak' else n_total * world + k * B)
            first = eng.store_rows('belief', rows)
            block_ids.append(np.arange(first, first + B, dtype=np.int32))
            del rows
    eng.set_beliefs(beliefs)
    shard = EngineShard(eng, m.gamma, carrier=torch.device('cpu') if (distributed and backend == 'gloo') else None)
    host = HostResults(eng, B)
    exchange_rows = exchange_rows_env

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    exchange_parts = []

    def step():
        if rotate:                                                    # a block the engine has no indexes of
            eng.select_beliefs(block_ids[step.n % len(block_ids)])
            step.n += 1
        if distributed:
            if exchange_rows:
                rows, count, idx, acts, keep, st = shard.run_resident_unique()
                gather_unique(dist, None, rows, count, idx, acts, keep, n_total)
                return st
            # local backup -> one all-gather of integers -> global dedup -> every replica appends the distinct rows
            eng.reset_store('alpha')                                # same store contents every step
            timing = {}
            st = sharded_engine_step(shard, dist, None, n_total, timing=timing)[5]
            exchange_parts.append(timing)
            return st
        if args.two_calls:
            st = eng.run(m.gamma)
            eng.fetch_compact_into(host.rows, host.index, host.actions)   # U rows + index + actions -> pinned host, synchronised
        else:
            # the same in one call, the rows leaving for the host while the refinement still runs (pbvi_backup_run_fetch)
            st, _, _ = eng.run_fetch_into(m.gamma, host.rows, host.slot, host.index, host.actions)
        return st
    step.n = 0

    per_step, elapsed, stats = timed_steps(step, args.steps, args.warmup, fence)
    if distributed:
        t = torch.tensor(np.concatenate([[elapsed], per_step]), dtype=torch.float64, device='cpu' if backend == 'gloo' else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # slowest rank, step by step and over the region
        elapsed, per_step = float(t[0].item()), t[1:].cpu().numpy()

    # device-resident variant: the same backups with the results left in HBM
    resident_ms = None
    host_ms = None
    reused = None
    if not distributed:
        if rotate:                                    # round 2's metric: one block backed up again and again
            eng.set_beliefs(beliefs)

            def reused_step():
                if args.two_calls:
                    eng.run(m.gamma)
                    eng.fetch_compact_into(host.rows, host.index, host.actions)
                else:
                    eng.run_fetch_into(m.gamma, host.rows, host.slot, host.index, host.actions)
            per_r, _, _ = timed_steps(reused_step, args.steps, 3, fence)
            reused = float(np.median(per_r))
        _, el2, _ = timed_steps(lambda: eng.run(m.gamma), args.steps, 2, fence)
        resident_ms = el2 / args.steps * 1e3
        # PCIe-inclusive variant: pageable host beliefs in, the expanded [B][S] alpha' matrix out to pageable memory
        # (the caller's arrays as a caller of an f32 engine would hold them: engine dtype, allocated once)
        b_host = np.ascontiguousarray(beliefs, dtype=eng.np_dtype)
        full = np.zeros((B, m.S), dtype=eng.np_dtype)
        def host_step():
            eng.set_beliefs(b_host)
            eng.run(m.gamma)
            eng.fetch()
            eng.fetch_full(full)
        host_step()                                  # the engine's pinned bounce buffer grows to the matrix's size once
        t1 = time.perf_counter()
        for _ in range(3):
            host_step()
        host_ms = (time.perf_counter() - t1) / 3 * 1e3
        del b_host, full

    if rank == 0:
        K = args.steps
        med = float(np.median(per_step))
        sparse = args.mode == 'sparse'
        out = {
            'metric': 'alpha-vector backups/sec', 'value': n_total / med, 'unit': 'backups/s',
            'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': med * 1e3, 'higher_is_better': True,
            'scaling': args.scaling, 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'olfactory-{m.S} {"reachable-sparse" if sparse else "dense-projection"} R={m.R} backup '
                                   f'(S={m.S}, A={m.A}, O={m.O}), V={args.alphas} alpha-vectors, '
                                   + (f'B={B} beliefs per GPU' if args.scaling == 'weak' else f'B={n_total} beliefs over {world} GPU(s)'),
                       'S': m.S, 'A': m.A, 'O': m.O, 'R': m.R, 'V': args.alphas, 'B_per_gpu': B, 'B_total': n_total,
                       'step': (('pbvi_beliefs_select (next block of this rank\'s belief store) + ' if rotate else '') +
                                'local backup + all-gather of ' + ('alpha rows' if exchange_rows else 'row keys')
                                + ' + global dedup + append of the distinct rows to every replica\'s alpha store') if distributed
                               else ('pbvi_beliefs_select (next block of the device belief store) + ' if rotate else '') +
                                    ('pbvi_backup_run + pbvi_backup_fetch_compact' if args.two_calls else 'pbvi_backup_run_fetch (= run + '
                                     'fetch_compact in one call; rows of the provisional decision cross PCIe under the refinement)') +
                                    ' -- U distinct rows, index, actions into pinned host memory, synchronised',
                       'parallelism': f'belief-sharded x{world}' if distributed else 'single GPU',
                       'resident_inputs': ('alpha set and the rows of the belief store in HBM before the timed region; every step '
                                           f'selects a block of {B} store rows it has not indexed (rotation over {len(block_ids)} blocks: '
                                           'sort order, zero-tile map, per-belief tile lists and dead-triple flags are rebuilt inside the '
                                           'step); nothing computed from the alpha set is kept between steps') if rotate else
                                          'alpha set and belief block in HBM before the timed region, with the indexes the engine '
                                          'builds of a belief block when it becomes resident / is first backed up (sort order, '
                                          'zero-tile map, dead-triple flags, per-belief tile lists); nothing computed from the '
                                          'alpha set is kept between steps'},
            'value_statistic': f'beliefs / median of the {K} timed steps (SURVEY 8d)',
            'value_mean_over_region': n_total * K / elapsed, 'ms_per_step_mean': elapsed / K * 1e3,
            'ms_per_step_min': float(per_step.min()) * 1e3, 'ms_per_step_max': float(per_step.max()) * 1e3,
            'roofline': gemm_roofline(stats, args.dtype, args.mode) if stats and stats[0] else None,
            # the executed MFMA work of the step at the fp32 peak / the device time of the whole step (all kernels)
            'step_frac_of_mfma_bound': (float(np.mean([s['score_flops_executed'] for s in stats])) / (PEAK_F32_MFMA_TFLOPS * 1e12)
                                        / (float(np.mean([s['ms_total'] for s in stats])) * 1e-3))
            if stats and stats[0] and args.dtype == 'f32' else None,
            'stage_ms': {k: float(np.mean([s[k] for s in stats])) for k in STAGES} if stats and stats[0] else None,
        }
        if stats and stats[-1]:
            st = stats[-1]
            out.update({'unique_rows': int(st['n_unique']), 'refined_pairs': int(st['n_refined']), 'dead_pairs': int(st['n_dead']),
                        'refined_actions': int(st['n_refined_actions']), 'pairs': int(st['n_pairs']), 'split_k': int(st['split_k'])})
        if exchange_parts:
            # host-side split of the sharded step on rank 0 (wall clock around each part; the timed steps only): local
            # backup + the engine packing its message | all-gather | message to host | merge | append of the distinct rows
            parts = exchange_parts[-K:]
            out['exchange_ms'] = {k: float(np.median([p[k] for p in parts])) for k in parts[0]}
            out['exchange_ms']['note'] = ('collective = ' + str(dist.get_backend()) + f', {world} rank(s); RCCL with more than one '
                                          'rank has not run on GPUs in any round (one-GPU boxes)')
        if reused is not None:
            out['value_reused_block'] = B / reused
            out['ms_per_step_reused_block'] = reused * 1e3
        if resident_ms is not None:
            out['value_device_resident'] = B / (resident_ms * 1e-3)
            out['ms_per_step_device_resident'] = resident_ms
        # HBM-side bytes per launch of the roofline kernel come from separate rocprofv3 --pmc passes (FETCH_SIZE /
        # WRITE_SIZE cannot be read inside this process); reported only for the exact workload they were measured on
        # (profiles/collect_r03.sh -> profiles/r03_pmc_traffic.json, one entry per configuration).
        if (sparse and args.dtype == 'f32' and m.S == 30000 and m.R == 1 and args.alphas == 1024 and B == 1024
                and out['roofline'] is not None):
            attach_traffic(out['roofline'], 'c4')
            sr = stage_rooflines(out['stage_ms'], 'c4')
            if sr:
                out['stage_roofline'] = sr
        if host_ms is not None:
            out['pcie_inclusive'] = {'ms_per_step': host_ms, 'value': B / (host_ms * 1e-3), 'unit': 'backups/s',
                                     'what': 'pageable beliefs (engine dtype) uploaded + run + unique rows and the expanded [B][S] alpha\' matrix fetched into a pageable array the caller keeps; mean of 3 after one warm-up'}
    host.close()
    eng.close()

    # The other BASELINE configurations that fit one GPU, in the same process (single-GPU default run only)
    default_workload = (world == 1 and not distributed and args.mode == 'sparse' and args.dtype == 'f32' and args.reach == 1
                        and args.grid == '75x400' and args.alphas == 1024 and B == 1024)
    if rank == 0 and args.secondary == 'auto' and default_workload:
        sec = {}
        try:
            sec['c4_unfused'] = measure_config('the headline workload with the Gamma projection as a kernel of its own (round 1\'s '
                                               'pipeline: k_project writes Gamma, the stream-K GEMM reads it)', m, alpha, beliefs,
                                               'f32', 'sparse', 20, 5, fence, fused=False)
            sec['c4_f64'] = measure_config('olfactory-30000 reachable-sparse R=1, fp64 engine (the reference\'s precision; scores '
                                           'screened in fp32, near-ties re-decided from the fp64 operands)',
                                           m, alpha, beliefs, 'f64', 'sparse', 10, 3, fence, pmc_tag='c4_f64')
            sec['c4_f64_pure'] = measure_config('olfactory-30000 reachable-sparse R=1, fp64 engine with the screen off '
                                                '(fp64 MFMA score GEMM)', m, alpha, beliefs, 'f64', 'sparse', 5, 2, fence,
                                                screen='off', pmc_tag='c4_f64_pure')
            m5 = synth.olfactory_model(H=H, W=W, R=5)
            alpha5, _ = synth.alpha_set(m5, args.alphas)
            beliefs5 = synth.belief_points(m5, B)
            sec['c4_r5'] = measure_config('olfactory-30000 reachable-sparse R=5 (stochastic moves)', m5, alpha5, beliefs5,
                                          'f32', 'sparse', 20, 5, fence, pmc_tag='c4_r5')
            del m5, alpha5, beliefs5
            # the protocol behind the reference's published 0.63 s per backup (Olfactory_Alternation_Paper_Wrap.ipynb:728-734):
            # ~100 new beliefs against a value function of thousands of rows, fp64 -- here B = 100, V = 8192, the engine
            # solve(use_gpu=True) uses by default (fp64, fp32 screen, belief-side formulation chosen by the cost model)
            alpha8k, _ = synth.alpha_set(m, 8192)
            sec['solve_shape'] = measure_config('olfactory-30000 R=1, the solve loop\'s shape: B=100 new beliefs x V=8192 alpha-vectors, '
                                                'fp64 engine as PBVI_Solver.solve(use_gpu=True) runs it', m, alpha8k, beliefs[:100],
                                                'f64', 'sparse', 20, 5, fence)
            del alpha8k
            # the shape at which the reference's CuPy path ran out of memory (Sea_Robin_Real.ipynb:913: 21.95 GB for Gamma in
            # fp64 at |V| = 1386, S = 61875, A = 16, O = 2); synthetic tables of that shape
            from types import SimpleNamespace
            S2, A2, O2, rs2, rto2, er2, alpha2, bel2 = synth.sea_robin_like()
            m2 = SimpleNamespace(S=S2, A=A2, O=O2, R=1, reachable_states=rs2, rto=rto2, expected_rewards=er2, gamma=0.99)
            sec['sea_robin_v1386'] = measure_config('sea-robin shape S=61875, A=16, O=2, R=1 at |V|=1386 (where the reference\'s '
                                                    'CuPy run died), B=100, fp64 engine; synthetic tables', m2, alpha2, bel2, 'f64',
                                                    'sparse', 10, 3, fence)
            sec['sea_robin_v1386_alpha_side_f32'] = measure_config('the same shape, fp32 engine, alpha-side formulation (Gamma for all '
                                                                   '1386 x 32 rows on the device)', m2, alpha2, bel2, 'f32', 'sparse', 10, 3,
                                                                   fence, formulation='alpha')
            del m2, rs2, rto2, er2, alpha2, bel2
            sec['c3_dense'] = measure_config('olfactory-30000 dense projection (|A||O| MFMA GEMMs over densified T.O, every '
                                             'tile of both GEMMs multiplied), fp32', m, alpha, beliefs, 'f32', 'dense', 3, 1,
                                             fence, true_dense=True, pmc_tag='c3_dense')
        except (MemoryError, RuntimeError) as e:          # a smaller card: report what was measured
            sec['error'] = f'{type(e).__name__}: {e}'
        out['secondary'] = sec
    if rank == 0:
        if world == 1 and not distributed and args.cpu_sample > 0:
            out['cpu_baseline'] = cpu_baseline(m, alpha, beliefs, min(args.cpu_sample, B))
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
