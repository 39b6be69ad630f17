import csv, glob, sys, collections
root = sys.argv[1]; pat = sys.argv[2]
f = glob.glob(root + '/**/*counter_collection.csv', recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
names = {}
for r in csv.DictReader(open(f)):
    if pat not in r['Kernel_Name']: continue
    d = int(r['Dispatch_Id']); names[d] = r['Kernel_Name'][:60]
    per[d][r['Counter_Name']] += float(r['Counter_Value'])
ds = sorted(per)
print(len(ds), 'dispatches of', pat)
for label, sel in (('all', ds), ('last 30', ds[-30:])):
    tot = collections.defaultdict(float)
    for d in sel:
        for k, v in per[d].items(): tot[k] += v
    print(label, {k: round(v / len(sel), 1) for k, v in tot.items()})
