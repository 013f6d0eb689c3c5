import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/*/*_counter_collection.csv')[0]
per=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('nb::','')
    per[(n,r['Counter_Name'])].append(float(r['Counter_Value']))
for (n,c),v in sorted(per.items()):
    if n.startswith(('k_leaf_sums','k_pyr_sub','k_keys','k_sort_scatter')):
        print('%-28s %-12s launches %4d  first10 mean %10.1f  last10 mean %10.1f' % (n[:28],c,len(v),sum(v[3:13])/10,sum(v[-10:])/10))
