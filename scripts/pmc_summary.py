import csv, collections, re, sys
for d in sys.argv[1:]:
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.defaultdict(float)
    seen=set()
    for r in csv.DictReader(open(f'{d}/pmc_counter_collection.csv')):
        n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); n=re.sub(r'^void ','',n).split('(')[0]
        agg[n][r['Counter_Name']]+=float(r['Counter_Value'])
        key=(r['Dispatch_Id'])
        if key not in seen:
            seen.add(key); cnt[n]+=1; dur[n]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    for n in sorted(agg, key=lambda n:-dur[n])[:8]:
        c=cnt[n]; a=agg[n]
        print(f'{d} {n}: calls {c} avg_us {dur[n]/c/1e3:.1f}')
        line=[]
        for k,v in a.items(): line.append(f'{k}={v/c:.3g}')
        print('     '+'  '.join(line))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in a and 'SQ_BUSY_CYCLES' in a:
            pass
