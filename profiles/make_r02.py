"""Turn gpurun_out/r02prof (collect_r02.sh) into the round-2 files kept under profiles/:
    r02_final_summary.md, r02_final_kernel_stats.csv, r02_pmc_traffic.json, r02_fsvi300_kernel_breakdown.md
Usage: python profiles/make_r02.py gpurun_out/r02prof"""
import csv
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def mfma_busy(path):
    per = {}
    with open(path, newline='') as fh:
        for r in csv.DictReader(fh):
            k = (r['Dispatch_Id'], r['Kernel_Name'])
            per.setdefault(k, {})
            per[k][r['Counter_Name']] = per[k].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    agg = {}
    for (_, name), c in per.items():
        a = agg.setdefault(name, {'n': 0})
        a['n'] += 1
        for kk, v in c.items():
            a[kk] = a.get(kk, 0.0) + v
    return agg


def stats_table(path, top=18):
    rows = list(csv.DictReader(open(path, newline='')))
    tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
    lines = ['| kernel | calls | total ms | avg us |', '|---|---|---|---|']
    for r in rows[:top]:
        lines.append(f"| `{r['Name'][:64]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.1f} | {float(r['AverageNs']) / 1e3:.1f} |")
    return tot, lines


def main():
    root = sys.argv[1]
    out = [open(os.path.join(root, 'summary', 'summary.md')).read().rstrip(), '']
    shutil.copy(os.path.join(root, 'summary', 'kernel_stats.csv'), os.path.join(HERE, 'r02_final_kernel_stats.csv'))
    shutil.copy(os.path.join(root, 'summary', 'pmc_traffic.json'), os.path.join(HERE, 'r02_pmc_traffic.json'))
    # agreement of the live figure with rocprofv3's durations of the same launches
    bench = json.load(open(os.path.join(root, 'bench_k.json')))
    live = bench['roofline']['ms_per_launch']
    launches = []
    hdr = open(os.path.join(root, 'kernel_trace_header.csv')).read().strip().replace('"', '').split(',')
    i0, i1 = hdr.index('Start_Timestamp'), hdr.index('End_Timestamp')
    for row in csv.reader(open(os.path.join(root, 'gemm_launches.csv'), newline='')):
        launches.append((int(row[i1]) - int(row[i0])) / 1e3)
    w, k = bench['warmup'], bench['steps']
    timed = launches[w:w + k]
    out += ['## Agreement of `bench.py`\'s live kernel time with rocprofv3 (same profiled run)', '',
            f'`bench.py` brackets the score GEMM with HIP events on the engine\'s stream: **{live:.4f} ms** per launch over its '
            f'{k} timed steps (after {w} warm-up steps).  rocprofv3\'s durations of those same {k} launches of '
            f'`{bench["roofline"]["kernel"].split(" ")[0]}` (launches {w + 1}..{w + k} of the process): '
            + ', '.join(f'{x:.0f}' for x in timed) + f' us, mean **{sum(timed) / len(timed) / 1e3:.4f} ms**.  '
            f'(The table above averages all {len(launches)} launches of the process: warm-up, timed, the device-resident '
            f'repeat and the 3 PCIe-inclusive steps.)  Unprofiled, the default `python bench.py` run is in '
            f'`r02_bench_default_run.json`.', '']
    agg = mfma_busy(glob.glob(os.path.join(root, 'prof_m', '*counter_collection.csv'))[0])
    out += ['## MFMA utilisation (`--pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES`, one pass)', '',
            'MFMA busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); kernels run ~10 % slower while '
            'counters are collected, the ratio is what matters.', '',
            '| kernel | launches | GRBM_GUI_ACTIVE | SQ_VALU_MFMA_BUSY_CYCLES | MFMA pipe busy |', '|---|---|---|---|---|']
    for name, a in agg.items():
        if 'gemm' in name and a.get('GRBM_GUI_ACTIVE'):
            n = a['n']
            g, mm = a['GRBM_GUI_ACTIVE'] / n, a['SQ_VALU_MFMA_BUSY_CYCLES'] / n
            out.append(f"| `{name[:60]}` | {n} | {g / 1e6:.2f} M | {mm / 1e6:.1f} M | **{(mm / 1024) / (g / 8) * 100:.1f} %** |")
    out.append('')
    open(os.path.join(HERE, 'r02_final_summary.md'), 'w').write('\n'.join(out) + '\n')
    # FSVI-300 kernel tables
    lines = ['# rocprofv3 --kernel-trace --stats of `examples/olfactory_fsvi.py --expansions 300 --growth 100` (round 2)', '']
    for tag, label in (('fsvi32', 'f32 engine'), ('fsvi64', 'f64 engine (the default of solve(use_gpu=True); backups screened in fp32 once large enough)')):
        log = open(os.path.join(root, f'{tag}.log')).read().strip().splitlines()
        wall = next((ln for ln in reversed(log) if 'wall=' in ln), '')
        tot, tab = stats_table(glob.glob(os.path.join(root, tag, '*kernel_stats.csv'))[0])
        lines += [f'## {label}', '', f'`{wall}`', '', f'total kernel time **{tot:.0f} ms**; the rest of the wall time is the host mirror '
                  'of the reference\'s containers (byte-key dedup, object walks) and launch / synchronisation overhead', ''] + tab + ['']
    open(os.path.join(HERE, 'r02_fsvi300_kernel_breakdown.md'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(out[-12:]))


if __name__ == '__main__':
    main()
