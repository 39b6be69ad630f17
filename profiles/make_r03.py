"""Turn gpurun_out/r03prof (profiles/collect_r03.sh) into the round-3 files kept under profiles/:
    r03_kernel_tables.md    per configuration: bench.py's line of the profiled run, the rocprofv3 --kernel-trace --stats table,
                            the dominant kernel's per-launch durations beside the live HIP-event figure, PMC traffic per kernel
    r03_pmc_traffic.json    per configuration: corrected HBM-side bytes per launch of every kernel (bench.py reads `traffic` here)
    r03_kernel_stats_<cfg>.csv, r03_fsvi300_{f32,f64}.log, r03_dist_1rank_rccl.json, r03_bench_default_run.json
Usage: python profiles/make_r03.py gpurun_out/r03prof"""
import csv
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CONFIGS = [('c4', 'headline: olfactory-30000 R=1, V=B=1024, f32, fresh belief block per step'),
           ('c4_r5', 'R=5 (stochastic moves), f32, block reused'),
           ('c4_f64', 'fp64 engine, fp32 screen, block reused'),
           ('c4_f64_pure', 'fp64 engine, PBVI_F64_SCREEN=off (fp64 MFMA GEMM), block reused'),
           ('c3_dense', 'dense projection, PBVI_GEMM_DENSE=1 (every tile of both GEMMs), f32, block reused')]


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    root = sys.argv[1]
    out = ['# Round 3: rocprofv3 evidence per benchmark configuration (MI355X)', '',
           'Collected by `profiles/collect_r03.sh` (one rocprofv3 pass per counter set, program directly behind `--`).  HBM-side '
           'bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, HBM section).  `live` = what `bench.py` measured with '
           'HIP events on the engine\'s stream in the SAME profiled process.', '']
    pmc_all = {}
    for tag, what in CONFIGS:
        d = os.path.join(root, tag)
        if not os.path.exists(os.path.join(d, 'bench_k.json')):
            continue
        b = last_json(os.path.join(d, 'bench_k.json'))
        shutil.copy(os.path.join(d, 'k_kernel_stats.csv'), os.path.join(HERE, f'r03_kernel_stats_{tag}.csv'))
        pmc = json.load(open(os.path.join(d, 'pmc.json')))
        pmc_all[tag] = pmc
        roof = b['roofline']
        out += [f'## {tag} -- {what}', '',
                f"`bench.py` in the profiled run: {b['value']:.0f} backups/s, {b['ms_per_step']:.3f} ms per step; dominant kernel "
                f"`{roof['kernel'].split(' ')[0]}`: live {roof['ms_per_launch']:.4f} ms per launch, {roof['achieved']:.1f} "
                f"{roof['unit']} = {roof['frac']:.3f} of {roof['peak']}.", '']
        rows = list(csv.DictReader(open(os.path.join(d, 'k_kernel_stats.csv'), newline='')))
        out += ['| kernel | calls | avg us | total ms | % |', '|---|---|---|---|---|']
        for r in rows[:16]:
            out.append(f"| `{r['Name'][:72]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                       f"{float(r['Percentage']):.2f} |")
        out.append('')
        # per-launch durations of the dominant kernel: the timed launches follow the warm-up ones
        lrows = list(csv.reader(open(os.path.join(d, 'launches.csv'), newline='')))
        if len(lrows) > 1:
            hdr = lrows[0]
            i0, i1 = hdr.index('Start_Timestamp'), hdr.index('End_Timestamp')
            dur = [(int(r[i1]) - int(r[i0])) / 1e3 for r in lrows[1:]]
            w, k = b['warmup'], b['steps']
            timed = dur[w:w + k]
            if timed:
                out += [f"rocprofv3 durations of launches {w + 1}..{w + k} of that kernel (the timed steps): "
                        + ', '.join(f'{x:.0f}' for x in timed) + f" us, mean **{sum(timed) / len(timed) / 1e3:.4f} ms** "
                        f"(live: **{roof['ms_per_launch']:.4f} ms**); all {len(dur)} launches of the process average "
                        f"{sum(dur) / len(dur) / 1e3:.4f} ms.", '']
        out += ['| kernel | launches | FETCH_SIZE KB | WRITE_SIZE KB | corrected MB per launch |', '|---|---|---|---|---|']
        for name, v in list(pmc['kernels'].items())[:10]:
            out.append(f"| `{name[:60]}` | {v['launches']} | {v['FETCH_SIZE_KB']:.0f} | {v['WRITE_SIZE_KB']:.0f} | {v['traffic_bytes'] / 1e6:.1f} |")
        if 'mfma' in pmc:
            m = pmc['mfma']
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
            clocks = m['GRBM_GUI_ACTIVE'] / 8.0
            busy = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (256 * 4) / clocks if clocks else 0.0
            out += ['', f"MFMA pipe of the dominant kernel: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = "
                        f"**{busy:.3f}** of the kernel's cycles; effective clock GRBM_GUI_ACTIVE / 8 / live duration = "
                        f"{clocks / (roof['ms_per_launch'] * 1e-3) / 1e9:.2f} GHz."]
        out.append('')
    with open(os.path.join(HERE, 'r03_kernel_tables.md'), 'w') as fh:
        fh.write('\n'.join(out) + '\n')
    json.dump({t: {'kernel': p['dominant_kernel'], 'traffic_bytes': p.get('traffic_bytes'), 'kernels': p['kernels'], 'method': p['method']}
               for t, p in pmc_all.items()}, open(os.path.join(HERE, 'r03_pmc_traffic.json'), 'w'), indent=1)
    for src, dst in (('fsvi300_f32.log', 'r03_fsvi300_f32.log'), ('fsvi300_f64.log', 'r03_fsvi300_f64.log'),
                     ('dist_1rank_rccl.json', 'r03_dist_1rank_rccl.json'), ('bench_default.json', 'r03_bench_default_run.json')):
        if os.path.exists(os.path.join(root, src)):
            shutil.copy(os.path.join(root, src), os.path.join(HERE, dst))
    print('\n'.join(out[:40]))


if __name__ == '__main__':
    main()
