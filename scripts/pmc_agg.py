#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc counter_collection CSV per kernel name: sum of every counter, dispatch count."""
import collections, csv, glob, os, re, sys
d = sys.argv[1]
files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for f in files:
  for r in csv.DictReader(open(f)):
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'\(.*', '', name).replace('void ', '')[:44]
    agg[name][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r.get('Dispatch_Id'), name)
    if key not in seen:
      seen.add(key)
      cnt[name] += 1
counters = sorted({c for v in agg.values() for c in v})
print('%-44s %6s ' % ('kernel', 'disp') + ' '.join('%16s' % c[:16] for c in counters))
for name in sorted(agg, key=lambda n: -agg[n].get(counters[0], 0)):
  print('%-44s %6d ' % (name, cnt[name]) + ' '.join('%16.4g' % agg[name].get(c, 0) for c in counters))
