import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
print('total kernel ms', round(tot, 1))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
    print(r['Name'][:78].ljust(78), r['Calls'].rjust(6), str(round(float(r['TotalDurationNs']) / 1e6, 1)).rjust(8), str(round(float(r['AverageNs']) / 1e3, 1)).rjust(8))
