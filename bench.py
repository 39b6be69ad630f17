"""Headline benchmark: alpha-vector backups / second at |S|=30000, |B|=1024 per GPU.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1)

A step is ONE point-based backup (``pbvi_backup_run``: Gamma projection, score GEMM, argmax +
fp64 tie refinement, action selection, alpha' assembly) of the resident belief block
against the resident alpha set -- inputs are in HBM when the timed region starts, results
stay in HBM.  With N>1 the beliefs are sharded (1024 per GPU, weak scaling) and each step
ends with the RCCL all-gather of the new alpha rows.  Workload: the synthetic olfactory
model of SURVEY.md 8d (BASELINE config "reachable-sparse", R=1 as in every large model of
the reference).  Prints ONE JSON line on rank 0.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--beliefs', type=int, default=1024, help='beliefs per GPU')
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
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N')
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or os.environ.get('PBVI_FORCE_DIST') == '1'   # the latter: rehearse the RCCL path on one GPU
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from pomdp_pbvi_exploration_amd import synth
    from pomdp_pbvi_exploration_amd.engine import Engine
    from pomdp_pbvi_exploration_amd.dist import EngineShard, gather_packed, gather_unique

    H, W = (int(x) for x in args.grid.split('x'))
    m = synth.olfactory_model(H=H, W=W, R=args.reach)
    alpha, _ = synth.alpha_set(m, args.alphas)
    B = args.beliefs
    beliefs = synth.belief_points(m, B, start=rank * B)            # this rank's block of the global set

    eng = Engine(m.S, m.A, m.O, m.R, m.reachable_states, m.rto, m.expected_rewards, dtype=args.dtype, device=local_rank,
                 mode=args.mode)
    eng.set_formulation(args.formulation)
    eng.set_alpha(alpha)
    eng.set_beliefs(beliefs)
    shard = EngineShard(eng, m.gamma)

    exchange_rows = os.environ.get('PBVI_EXCHANGE') == 'rows'       # A-B only

    def step():
        if distributed:
            # local backup, then ONE all-gather of integers: per-belief index / action / keep and the keys
            # (a*, v*[a*, :]) of this rank's distinct alpha' rows; every rank rebuilds all rows from the keys against
            # its replica of the alpha set (dist.gather_packed).  PBVI_EXCHANGE=rows moves the rows themselves instead.
            if exchange_rows:
                rows, count, idx, acts, keep, st = shard.run_resident_unique()
                gather_unique(dist, None, rows, count, idx, acts, keep, B * world)
            else:
                meta, per, kw, st = shard.run_resident_packed()
                gather_packed(dist, None, meta, per, kw, B * world, shard.assemble)
            return st
        return eng.run(m.gamma)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # PCIe-inclusive variant (host beliefs in, host alpha' out) -- reported beside, never as `value`
    host_ms = None
    if not distributed:
        t1 = time.perf_counter()
        for _ in range(3):
            eng.set_beliefs(beliefs)
            eng.run(m.gamma)
            eng.fetch()
        host_ms = (time.perf_counter() - t1) / 3 * 1e3

    if rank == 0:
        K = args.steps
        ms_step = elapsed / K * 1e3
        ms_score = float(np.mean([s['ms_score'] for s in stats]))
        flops_dense = stats[0]['score_flops']                   # 2*B*S*A*O*V
        flops = stats[0]['score_flops_executed']                # same, restricted to structurally non-zero tiles
        achieved = flops / (ms_score * 1e-3) / 1e12
        peak = PEAK_F32_MFMA_TFLOPS if args.dtype == 'f32' else PEAK_F64_MFMA_TFLOPS
        gemm_name = ('k_gemm_nt_f32_streamk' if args.dtype == 'f32' else 'k_gemm_nt_f64_mfma')
        out = {
            'metric': 'alpha-vector backups/sec', 'value': B * world * K / elapsed, 'unit': 'backups/s',
            'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': ms_step, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'olfactory-{m.S} {"reachable-sparse" if args.mode == "sparse" else "dense-projection"} R={m.R} backup (S={m.S}, A={m.A}, O={m.O}), '
                                   f'V={args.alphas} alpha-vectors, B={B} beliefs per GPU',
                       'S': m.S, 'A': m.A, 'O': m.O, 'R': m.R, 'V': args.alphas, 'B_per_gpu': B,
                       'parallelism': (f'belief-sharded x{world}, 1 all-gather of ' + ('alpha rows' if exchange_rows else 'row keys (rows rebuilt per rank)')) if distributed else 'single GPU'},
            'roofline': {'bound': 'mfma', 'kernel': f'{gemm_name} (belief x Gamma score GEMM, non-zero tiles)',
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': None,
                         'flops_per_launch': flops, 'ms_per_launch': ms_score,
                         'dense_flops_per_launch': flops_dense,
                         'dense_equivalent_tflops': flops_dense / (ms_score * 1e-3) / 1e12,
                         'tiles_run_over_dense': stats[0]['score_tiles_run'] / max(1, stats[0]['score_tiles_dense'])},
            'stage_ms': {k: float(np.mean([s[k] for s in stats])) for k in
                         ('ms_total', 'ms_project', 'ms_score', 'ms_argmax', 'ms_refine', 'ms_action', 'ms_assemble')},
            'unique_rows': int(stats[-1]['n_unique']), 'refined_pairs': int(stats[-1]['n_refined']), 'dead_pairs': int(stats[-1]['n_dead']),
            'refined_actions': int(stats[-1]['n_refined_actions']), 'pairs': int(stats[-1]['n_pairs']), 'split_k': int(stats[-1]['split_k']),
        }
        # HBM-side bytes per launch of the roofline kernel come from a separate rocprofv3 --pmc run
        # (FETCH_SIZE / WRITE_SIZE cannot be read inside this process); reported only for the exact
        # workload they were measured on.
        try:
            with open(os.path.join(REPO, 'profiles', 'r01_pmc_traffic.json')) as fh:
                pmc = json.load(fh)
            if (args.mode == 'sparse' and args.dtype == 'f32' and m.S == 30000 and m.R == 1 and args.alphas == 1024 and B == 1024):
                out['roofline']['traffic'] = pmc['traffic_bytes']
                out['roofline']['traffic_source'] = 'profiles/r01_pmc_traffic.json (rocprofv3 --pmc, separate passes)'
                # the HBM-bound stages beside the GEMM: PMC bytes of the stage's main kernel / the stage's live time
                # (the stage time also holds its small helper kernels, so these fractions are lower bounds)
                other = pmc.get('other_kernels', {})
                sec = {}
                for stage, kern in (('ms_project', 'k_project'), ('ms_argmax', 'k_argmax'), ('ms_refine', 'k_refine')):
                    if kern in other:
                        ms = out['stage_ms'][stage]
                        gbs = other[kern]['traffic_bytes'] / (ms * 1e-3) / 1e9
                        sec[kern] = {'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                     'frac': gbs / PEAK_HBM_GBS, 'traffic': other[kern]['traffic_bytes'], 'ms': ms}
                if sec:
                    out['stage_roofline'] = sec
        except (OSError, KeyError, ValueError):
            pass
        if args.mode == 'dense':   # the projection GEMMs dominate: report them as the roofline kernel
            ms_proj = float(np.mean([s['ms_project'] for s in stats]))
            pf, pfe = stats[0]['project_flops'], stats[0]['project_flops_executed']
            out['roofline'] = {'bound': 'mfma', 'kernel': 'k_gemm_nt_f32_mfma (dense projection, batched over (a,o))',
                               'achieved': pfe / (ms_proj * 1e-3) / 1e12, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': pfe / (ms_proj * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 'traffic': None,
                               'flops_per_launch': pfe, 'ms_per_launch': ms_proj, 'dense_flops_per_launch': pf,
                               'note': 'ms_project also holds the memset / scale-copy passes around the GEMM'}
        if host_ms is not None:
            out['pcie_inclusive'] = {'ms_per_step': host_ms, 'value': B / (host_ms * 1e-3), 'unit': 'backups/s'}
        if world == 1 and args.cpu_sample > 0:
            out['cpu_baseline'] = cpu_baseline(m, alpha, beliefs, min(args.cpu_sample, B))
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
