import csv,sys,glob,collections
for d in sys.argv[1:]:
    f=glob.glob(d+'/**/*kernel_trace.csv',recursive=True)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][-40:]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    print(d)
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:6]:
        v2=v[len(v)//2:]
        print('   %-42s n=%d avg(last half)=%.1f us min=%.1f'%(k,len(v),sum(v2)/len(v2),min(v)))
