"""Turn rocprofv3 CSV output into the summaries kept under profiles/.

Run on the GPU box after (each rocprofv3 call in its own pass, as gpurun requires):
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -o k -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_f -o f -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_w -o w -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0
    python3 profiles/summarize.py gpurun_out gpurun_out/summary
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE/WRITE_SIZE are KB;
FETCH_SIZE under-reports 16-byte-per-lane streams 2x on gfx950; Infinity-Cache hits are counted).
"""
import csv
import glob
import re
import json
import os
import sys


def find(root, pattern):
    hits = sorted(glob.glob(os.path.join(root, '**', pattern), recursive=True))
    return hits[0] if hits else None


def kernel_stats(path):
    rows = []
    with open(path, newline='') as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
    return rows


def counter_means(path, counter):
    """kernel name -> (launches, mean counter value per launch); values of one dispatch are summed over agents/dims."""
    per_dispatch = {}
    with open(path, newline='') as fh:
        for r in csv.DictReader(fh):
            if r.get('Counter_Name') != counter:
                continue
            key = (r.get('Dispatch_Id') or r.get('Correlation_Id'), r['Kernel_Name'])
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r['Counter_Value'])
    agg = {}
    for (_, name), v in per_dispatch.items():
        n, tot = agg.get(name, (0, 0.0))
        agg[name] = (n + 1, tot + v)
    return {k: (n, tot / n) for k, (n, tot) in agg.items()}


def main():
    root, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    lines = ['# Kernel profile (MI355X, rocprofv3)', '',
             'Workload: `python3 bench.py --steps 10 --warmup 10 --cpu-sample 0 --secondary none` (olfactory-30000 '
             'reachable-sparse R=1, V=1024, B=1024, f32 engine; every step = pbvi_backup_run + pbvi_backup_fetch_compact).', '']
    ks = find(os.path.join(root, 'prof_k'), '*kernel_stats.csv')
    if ks:
        rows = kernel_stats(ks)
        with open(os.path.join(out, 'kernel_stats.csv'), 'w') as fh:
            fh.write(open(ks).read())
        lines += ['## `rocprofv3 --kernel-trace --stats`', '', '| kernel | calls | avg us | total ms | % |', '|---|---|---|---|---|']
        for r in rows[:24]:
            lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                         f"{float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |")
        lines.append('')
    f = find(os.path.join(root, 'prof_f'), '*counter_collection.csv')
    w = find(os.path.join(root, 'prof_w'), '*counter_collection.csv')
    if f and w:
        fm, wm = counter_means(f, 'FETCH_SIZE'), counter_means(w, 'WRITE_SIZE')
        lines += ['## HBM-side traffic per launch (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate passes)', '',
                  'corrected bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024', '',
                  '| kernel | launches | FETCH_SIZE KB | WRITE_SIZE KB | corrected MB |', '|---|---|---|---|---|']
        table = []
        for name, (n, fv) in fm.items():
            wv = wm.get(name, (0, 0.0))[1]
            table.append((2 * fv + wv, name, n, fv, wv))
        for tot, name, n, fv, wv in sorted(table, reverse=True)[:10]:
            lines.append(f'| `{name[:60]}` | {n} | {fv:.1f} | {wv:.1f} | {tot * 1024 / 1e6:.1f} |')
        gem = sorted((t for t in table if 'k_gemm_nt_f32_streamk' in t[1]), reverse=True)   # the one that moved the most
        if gem:
            tot, name, n, fv, wv = gem[0]
            with open(os.path.join(out, 'pmc_traffic.json'), 'w') as fh:
                json.dump({'kernel': re.sub(r'^void ', '', name).split('(')[0].replace('pbvi::', ''),
                           'workload': 'olfactory-30000 reachable-sparse R=1 V=1024 B=1024 f32',
                           'launches_averaged': n, 'FETCH_SIZE_KB': fv, 'WRITE_SIZE_KB': wv,
                           'traffic_bytes': tot * 1024,
                           'other_kernels': {re.sub(r'^void ', '', nm).split('(')[0].split('<')[0].replace('pbvi::', ''):
                                             {'launches': nn, 'traffic_bytes': tt * 1024}
                                             for tt, nm, nn, _, _ in sorted(table, reverse=True)[:10]
                                             if 'pbvi::' in nm and nm != name},
                           'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bytes = '
                                     '(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md section HBM'}, fh, indent=1)
        lines.append('')
    with open(os.path.join(out, 'summary.md'), 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:40]))


if __name__ == '__main__':
    main()
