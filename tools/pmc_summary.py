import csv, sys, collections, glob
for d in sys.argv[1:]:
    f = glob.glob(d+'/*/*counter_collection.csv')[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void neb::','').replace('(neb::AtrousArgs)','').replace('(neb::TemporalArgs)','')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        acc[k]['_lds'] = [float(r['LDS_Block_Size'])]; acc[k]['_vgpr']=[float(r['VGPR_Count'])]
    for k,v in acc.items():
        if 'rocclr' in k: continue
        print(k, ' '.join(f"{c}={sum(x)/len(x):.4g}" for c,x in sorted(v.items())))
