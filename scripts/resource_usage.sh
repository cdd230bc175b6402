#!/bin/bash
# Prints per-kernel register / scratch / occupancy figures of every HIP source (no GPU needed).
cd "$(dirname "$0")/../automl_amd/csrc"
for f in ${@:-*.hip}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/ru_$$.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m:
        cur={'name':subprocess.run(['/usr/bin/c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()}; rows.append(cur); continue
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)'),('sgpr',r'TotalSGPRs: (\d+)')):
        m=re.search(pat,line)
        if m and cur is not None: cur[k]=m.group(1)
for r in rows:
    n=re.sub(r'\(anonymous namespace\)::','',r['name']); n=re.sub(r'\(.*','',n).replace('void ','')
    print('%-48s vgpr=%3s agpr=%3s sgpr=%3s scratch=%5s occ=%s lds=%s'%(n[:48],r.get('vgpr'),r.get('agpr'),r.get('sgpr'),r.get('scratch'),r.get('occ'),r.get('lds')))
"
done
rm -f /tmp/ru_$$.o
