import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'wino8' not in k: continue
    k='spatial' if 'wino8s' in k else 'linear'
    d[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in d:
    print(k, len(n[k]), {c: round(v/len(n[k])/1e6,2) for c,v in d[k].items()})
