import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.defaultdict(lambda: [0, 0.0])
prev = None
for i, r in enumerate(rows):
    n = r['Kernel_Name']
    if 'k_gemm_nt_f64_mfma<double>' in n or 'k_gemm_nt_f32_streamk' in n:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        g = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
        gy = int(r['Grid_Size_Y']) ; gz = int(r['Grid_Size_Z'])
        # what follows tells who called it
        nxt = ''
        for j in range(i + 1, min(i + 6, len(rows))):
            m = rows[j]['Kernel_Name']
            if 'k_argmax' in m or 'k_extra' in m: nxt = m.split('(')[0][-40:]; break
        key = (n.split('(')[0][-38:], 'blocks~%d' % (10 ** len(str(g * gy * gz))), nxt)
        agg[key][0] += 1; agg[key][1] += d
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(k, v[0], 'launches', round(v[1], 1), 'ms', round(v[1] / v[0], 3), 'ms each')
