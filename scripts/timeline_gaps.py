import csv,gzip,collections,re,sys
rows=list(csv.DictReader(gzip.open(sys.argv[1],'rt')))
for r in rows: r['start']=int(r['start']); r['end']=int(r['end'])
rows.sort(key=lambda r:r['start'])
stems=[i for i,r in enumerate(rows) if 'k_stem_fwd' in r['name']]
a,b=stems[-2],stems[-1]
print('step wall ms', (rows[b]['start']-rows[a]['start'])/1e6)
def short(n):
    n=n.replace('void ','').replace('(anonymous namespace)::','')
    n=re.sub(r'\(.*','',n)
    return n[:60]
for q in sorted(set(r['queue'] for r in rows[a:b])):
  step=[r for r in rows[a:b] if r['queue']==q]
  n=collections.Counter(); g=collections.Counter(); tot=collections.Counter()
  for p,r in zip(step,step[1:]):
    k=short(r['name']); n[k]+=1
    if r['start']-p['end']>1000: g[k]+=1; tot[k]+=r['start']-p['end']
  print('queue',q,'kernels',len(step),'busy ms %.2f'%(sum(r['end']-r['start'] for r in step)/1e6),'gap ms %.2f'%(sum(tot.values())/1e6), 'gaps', sum(g.values()))
  for k in sorted(n, key=lambda k:-tot[k])[:int(sys.argv[2]) if len(sys.argv)>2 else 12]:
    if g[k]: print('  %-62s %3d/%3d  %.1f us'%(k,g[k],n[k],tot[k]/1e3))
