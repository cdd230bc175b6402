#!/usr/bin/env python3
"""Instruction mix of one kernel from the gfx950 assembly of a HIP source (no GPU needed).
usage: isa_mix.py file.hip kernel_substring [top_n]"""
import collections, os, re, subprocess, sys, tempfile
src, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'automl_amd', 'csrc')
d = tempfile.mkdtemp()
base = os.path.splitext(os.path.basename(src))[0]
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-save-temps=obj',
                '-c', os.path.join(here, src), '-o', os.path.join(d, base + '.o')], capture_output=True, cwd=d)
s = open(os.path.join(d, base + '-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
  name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
  if pat not in name:
    continue
  body = m.group(2)
  c = collections.Counter(re.findall(r'^\s+([a-z_0-9]+)', body, re.M))
  print('==', name[:120], 'total instr', sum(c.values()))
  print('  ' + ', '.join('%s %d' % kv for kv in c.most_common(top)))
