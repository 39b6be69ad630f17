import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.defaultdict(lambda: [0, 0.0])
series = []
for i, r in enumerate(rows):
    n = r['Kernel_Name']
    if 'k_gemm_nt_f32_streamk' in n:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        prev = ''
        for j in range(i - 1, max(i - 12, 0), -1):
            m = rows[j]['Kernel_Name']
            if 'k_push_project' in m: prev = 'backup(push)'; break
            if 'k_project' in m: prev = 'backup(pull)'; break
            if 'k_argmax' in m or 'k_gemm' in m: break
        key = prev or 'value_max'
        agg[key][0] += 1; agg[key][1] += d
        series.append((key, round(d, 2)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(k, v[0], 'launches', round(v[1], 1), 'ms', round(v[1] / v[0], 3), 'ms each')
b = [d for k, d in series if k.startswith('backup')]
print('backup GEMM ms at expansions 50,100,150,200,250,299:', [b[i] for i in (50, 100, 150, 200, 250, min(299, len(b) - 1))])
vm = [d for k, d in series if k == 'value_max']
print('value_max GEMM ms sample:', vm[100::100])
