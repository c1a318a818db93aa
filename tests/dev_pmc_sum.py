import csv, sys, glob, collections
def load(d):
    f=glob.glob(d+'/runc/*counter_collection.csv')
    if not f: print("no file",d); return
    rows=list(csv.DictReader(open(f[0])))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in rows:
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
    for k in agg:
        if 'gnxr' in k: print(k, "dispatches",len(cnt[k]), {c:"%.4g"%v for c,v in agg[k].items()})
for d in sys.argv[1:]: print("==",d); load(d)
