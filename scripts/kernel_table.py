#!/usr/bin/env python3
"""Per-kernel table of a round's evidence: ms per step (rocprofv3 kernel statistics) next to the HBM traffic per step
(PMC passes, profiles/<tag>_traffic.json) and the rate the two give.  usage: kernel_table.py TAG [ROWS]"""
import csv, json, re, sys
tag = sys.argv[1]; nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = list(csv.DictReader(open('profiles/%s_kernel_stats_b128.csv' % tag)))
steps = [int(r['Calls']) for r in rows if 'k_stem_fwd_mfma' in r['Name']][0]
t = json.load(open('profiles/%s_traffic.json' % tag)); S = t['steps_in_run']; ks = t['kernels']
def short(n):
  n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
  return n.split('(')[0]
tot = 0; ttr = 0; out = []
for r in rows:
  n = short(r['Name']); ms = int(r['TotalDurationNs']) / 1e6 / steps; tot += ms
  key = [k for k in ks if k.strip() == n[:len(k.strip())] and len(k.strip()) >= min(len(n), 44)]
  tr = None
  if key:
    k = ks[key[0]]; tr = (k['fetch_bytes'] + k['write_bytes']) / S / 1e9; ttr += tr
  out.append((ms, n, int(r['Calls']) / steps, tr))
print('steps in the trace %d, kernel time %.2f ms/step, traffic %.1f GB/step' % (steps, tot, ttr))
for ms, n, c, tr in out[:nrows]:
  print('%7.3f ms %5.1f x  %-66s %s' % (ms, c, n[:66], '' if tr is None else '%6.2f GB %5.2f TB/s' % (tr, tr / ms)))
