"""Aggregate the --pmc passes of one profiled configuration (collect_r03.sh) into <dir>/pmc.json:
per kernel the launches, mean FETCH_SIZE / WRITE_SIZE (KB) and corrected bytes (2 FETCH + WRITE) x 1024
(MI355X_MICROARCH.md, HBM section), and for the dominant kernel the MFMA-busy share of its cycles."""
import csv, glob, json, os, re, sys

root, pat = sys.argv[1], sys.argv[2]


def means(sub, counters):
    hits = glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True)
    if not hits:
        return {}
    per = {}
    for r in csv.DictReader(open(hits[0], newline='')):
        if r['Counter_Name'] not in counters:
            continue
        k = (r['Dispatch_Id'], r['Kernel_Name'], r['Counter_Name'])
        per[k] = per.get(k, 0.0) + float(r['Counter_Value'])
    agg = {}
    for (_, name, c), v in per.items():
        n, tot = agg.get((name, c), (0, 0.0))
        agg[(name, c)] = (n + 1, tot + v)
    return {k: (n, tot / n) for k, (n, tot) in agg.items()}


def short(name):
    return re.sub(r'^void ', '', name).split('(')[0].replace('pbvi::', '')


f, w = means('f', {'FETCH_SIZE'}), means('w', {'WRITE_SIZE'})
m = means('m', {'GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES'})
kernels = {}
for (name, _), (n, fv) in f.items():
    wv = w.get((name, 'WRITE_SIZE'), (0, 0.0))[1]
    kernels[short(name)] = {'launches': n, 'FETCH_SIZE_KB': fv, 'WRITE_SIZE_KB': wv, 'traffic_bytes': (2 * fv + wv) * 1024}
dom = [k for k in kernels if pat in k]
out = {'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 '
                 '(MI355X_MICROARCH.md, HBM: KB units, 16-byte streams counted at half, Infinity-Cache hits included)',
       'dominant_kernel': max(dom, key=lambda k: kernels[k]['traffic_bytes']) if dom else None,
       'kernels': dict(sorted(kernels.items(), key=lambda kv: -kv[1]['traffic_bytes'])[:14])}
if out['dominant_kernel']:
    full = [nm for (nm, c) in m if pat in nm and c == 'GRBM_GUI_ACTIVE']
    if full:
        nm = max(full, key=lambda x: m[(x, 'GRBM_GUI_ACTIVE')][1])
        g, b, s = (m.get((nm, c), (0, 0.0))[1] for c in ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES'))
        out['mfma'] = {'GRBM_GUI_ACTIVE': g, 'SQ_VALU_MFMA_BUSY_CYCLES': b, 'SQ_BUSY_CYCLES': s}
    out['traffic_bytes'] = kernels[out['dominant_kernel']]['traffic_bytes']
json.dump(out, open(os.path.join(root, 'pmc.json'), 'w'), indent=1)
print(json.dumps(out, indent=1)[:1500])
