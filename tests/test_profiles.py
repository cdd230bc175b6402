"""The committed measurement evidence is internally consistent (no GPU needed): the bench line of the tree, the rocprofv3
kernel statistics profiles/CURRENT names, the PMC traffic file the line quotes and the kernel timeline agree with each other
and with what DESIGN.md says about them."""
import csv
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')


def _current():
  cur = os.path.join(PROF, 'CURRENT')
  if not os.path.exists(cur):
    pytest.skip('profiles/CURRENT not present')
  name = open(cur).read().strip()
  return name[:-len('_kernel_stats_b128.csv')], os.path.join(PROF, name)


def test_bench_line_kernel_statistics_and_traffic_of_the_current_tree_agree():
  tag, stats_path = _current()
  rows = list(csv.DictReader(open(stats_path)))
  line = json.loads(open(os.path.join(PROF, tag + '_bench_b128.json')).read())
  roof = line['roofline']
  # the contract fields of the line
  assert line['unit'] == 'images/sec' and line['n_gpus'] == 1 and line['higher_is_better'] is True
  assert line['dtype'] == 'bf16' and line['data'] == 'synthetic' and line['vs_baseline'] is None
  assert abs(line['value'] - 128 * 1e3 / line['ms_per_step']) <= 1e-6 * line['value']
  assert roof['bound'] == 'hbm' and roof['peak'] == 8000.0 and abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-9
  assert line['cpu_baseline']['kind'] == 'port' and line['cpu_baseline']['value'] > 0
  # achieved = algorithmic bytes per launch / average launch duration (both in the line)
  want = roof['algorithmic_bytes_per_launch'] / (roof['avg_launch_ms'] * 1e-3) / 1e9
  assert abs(want - roof['achieved']) <= 1e-3 * roof['achieved']
  # the rocprofv3 statistics of the same command: the kernels of the dominant entry point take the time the line says
  # (the profiled run is 10 steps: 3 timed replays + warm-up / profiled / roofline steps; launches per step from the table)
  mine = [r for r in rows if '::k_' in r['Name'] or r['Name'].startswith('k_') or r['Name'].startswith('void k_')]
  assert len(mine) >= 20
  dom = [r for r in rows if any(t in r['Name'] for t in ('pwt::k_pw_bwd_tile', 'pws::k_pw_bwd_fused', 'pwb::k_big_wgrad_bal',
                                                           'pwb::k_big_gemm<true', 'pws::k_pw_wgrad', 'pws::k_pw_dgrad',
                                                           'pws::k_noy_apply', 'pwt::k_gate_finish'))]
  stem = [r for r in rows if 'k_stem_fwd' in r['Name']]
  steps = int(stem[0]['Calls'])
  ms_per_step = sum(float(r['TotalDurationNs']) for r in dom) / steps / 1e6
  assert 0.8 * roof['per_kernel_ms']['edet_pw_bwd'] <= ms_per_step <= 1.1 * roof['per_kernel_ms']['edet_pw_bwd'], ms_per_step
  # PMC traffic of the dominant entry point: what the line quotes is what the traffic file of the same tag gives
  assert roof['traffic_source'] == tag + '_traffic.json'
  assert 0.5 <= roof['traffic'] / roof['algorithmic_bytes_per_launch'] <= 1.15      # r03 verdict: <= 1.15x


def test_timeline_of_the_replayed_step_has_no_idle_time():
  """in a graph replay the union of the kernel intervals of both queues covers the step."""
  tag, _ = _current()
  trace = os.path.join(PROF, tag + '_timeline_b128.csv.gz')
  if not os.path.exists(trace):
    pytest.skip('no timeline committed for %s' % tag)
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'timeline_gaps.py'), trace, '3'],
                       capture_output=True, text=True, check=True).stdout
  busy = [l for l in out.splitlines() if l.startswith('union of the kernel intervals')][0]
  idle_ms = float(busy.split(' ms busy, ')[1].split(' ms idle')[0])
  wall = [l for l in out.splitlines() if l.startswith('analysed:')][0]
  wall_ms = float(wall.split(', ')[1].split(' ms')[0])
  assert idle_ms <= 0.005 * wall_ms, out
  line = json.loads(open(os.path.join(PROF, tag + '_bench_b128.json')).read())
  assert abs(wall_ms - line['ms_per_step']) <= 0.03 * line['ms_per_step'], (wall_ms, line['ms_per_step'])
