import csv, collections, sys
path=sys.argv[1]
rows=list(csv.DictReader(open(path)))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.Counter()
seen=set()
for r in rows:
    k=r['Kernel_Name'].split('(')[0][-40:]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    key=(r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); cnt[k]+=1
for k,v in agg.items():
    if 'wave' not in k and 'trace' not in k and 'raygen' not in k: continue
    print(k, 'dispatches', cnt[k])
    for c,val in sorted(v.items()): print('   %-28s %.4g'%(c,val))
    if 'SQ_ACTIVE_INST_VALU' in v and v['SQ_ACTIVE_INST_VALU']:
        print('   lane utilization  %.1f%%' % (100*v['SQ_THREAD_CYCLES_VALU']/(v['SQ_ACTIVE_INST_VALU']*64)))
    if 'SQ_WAVE_CYCLES' in v and v.get('SQ_INSTS_VALU'):
        print('   wave-cycles(quad) per VALU inst %.2f' % (v['SQ_WAVE_CYCLES']/v['SQ_INSTS_VALU']))
