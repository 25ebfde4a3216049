import csv, sys, collections, glob
def load(d):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-60:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r['Dispatch_Id'])
        if key not in seen:
            seen.add(key); cnt[k] += 1
    return agg, cnt
for d in sys.argv[1:]:
    agg, cnt = load(d)
    for k, v in agg.items():
        if 'gemm128_kernel<float' in k or 'kfill_kernel<float, float' in k or 'double, true' in k:
            print(d, k, cnt[k], {a: '%.4g' % (b / cnt[k]) for a, b in v.items()})
