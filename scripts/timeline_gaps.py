"""Kernel timeline of one REPLAYED step out of a scripts/gpu_trace.sh trace: per queue the busy time and the holes between
consecutive kernels (attributed to the kernel that starts after the hole), and the union of the busy intervals of all queues.

usage: python scripts/timeline_gaps.py gpurun_out/TAG_trace.csv.gz [rows_per_queue]

A bench.py run is: eager / capture / replay warm-up steps, one eager fully profiled step, the timed replays, then as many
EAGER steps with HIP events around the dominant entry point (the roofline leg).  The steps are delimited by the stem kernel;
the shortest one is a replay (the eager legs carry host gaps and event barriers) and is the one analysed."""
import collections
import csv
import gzip
import re
import sys


def short(n):
  n = n.replace('void ', '').replace('(anonymous namespace)::', '')
  return re.sub(r'\(.*', '', n)[:60]


def main():
  rows = list(csv.DictReader(gzip.open(sys.argv[1], 'rt')))
  top = int(sys.argv[2]) if len(sys.argv) > 2 else 10
  for r in rows:
    r['start'], r['end'] = int(r['start']), int(r['end'])
  rows.sort(key=lambda r: r['start'])
  stems = [i for i, r in enumerate(rows) if 'k_stem_fwd' in r['name']]
  walls = [(rows[stems[i + 1]]['start'] - rows[stems[i]]['start'], i) for i in range(len(stems) - 1)]
  wall, si = min(walls)
  a, b = stems[si], stems[si + 1]
  step = rows[a:b]
  print('steps in the trace (ms): %s' % ' '.join('%.2f' % (w / 1e6) for w, _ in walls))
  print('analysed: step %d, %.3f ms from stem kernel to stem kernel, %d kernels' % (si, wall / 1e6, len(step)))
  iv = sorted((r['start'], r['end']) for r in step)
  busy, (cs, ce) = 0, iv[0]
  for s, e in iv[1:]:
    if s > ce:
      busy += ce - cs
      cs, ce = s, e
    else:
      ce = max(ce, e)
  busy += ce - cs
  print('union of the kernel intervals of all queues: %.3f ms busy, %.3f ms idle' % (busy / 1e6, (wall - busy) / 1e6))
  for q in sorted(set(r['queue'] for r in step)):
    ks = [r for r in step if r['queue'] == q]
    n, g, tot = collections.Counter(), collections.Counter(), collections.Counter()
    for p, r in zip(ks, ks[1:]):
      k = short(r['name'])
      n[k] += 1
      if r['start'] - p['end'] > 1000:
        g[k] += 1
        tot[k] += r['start'] - p['end']
    small = [r for r in ks if r['end'] - r['start'] < 15000]
    print('queue %s: %d kernels, %.2f ms busy, %d holes > 1 us (%.2f ms); %d kernels under 15 us = %.2f ms' % (
        q, len(ks), sum(r['end'] - r['start'] for r in ks) / 1e6, sum(g.values()), sum(tot.values()) / 1e6, len(small),
        sum(r['end'] - r['start'] for r in small) / 1e6))
    for k in sorted(n, key=lambda k: -tot[k])[:top]:
      if g[k]:
        print('    %-62s %3d of %3d launches after a hole, %.1f us' % (k, g[k], n[k], tot[k] / 1e3))


if __name__ == '__main__':
  main()
