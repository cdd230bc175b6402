#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; aggregated by scripts/pmc_agg.py) into
profiles/<tag>_traffic.json: HBM bytes per kernel name over the profiled run.

Units / corrections (MI355X_MICROARCH.md, HBM section, and the calibration below): both counters are in
KiB; on gfx950 FETCH_SIZE tallies each 128-byte fabric read request as 64 bytes, so it is doubled;
WRITE_SIZE is used as is.  Calibration on kernels of this library with exactly known traffic (r01e run,
3 steps of D0 640x640 batch 128): k_bn_res reads 1.65 GB/step -> FETCH_SIZE*2 = 1.65 GB, writes
1.21 GB/step -> WRITE_SIZE = 1.21 GB; k_focal writes 1.77 GB/step of dlogits -> WRITE_SIZE = 1.84 GB,
reads 1.81 GB -> FETCH_SIZE*2 = 2.06 GB; k_stem_fwd_mfma writes 839 MB -> WRITE_SIZE = 839 MB.

usage: pmc_traffic.py FETCH_AGG.txt WRITE_AGG.txt STEPS_IN_RUN OUT.json
"""
import json
import sys


def read(path):
  out = {}
  for line in open(path).read().splitlines()[1:]:
    parts = line.rsplit(None, 2)
    if len(parts) == 3:
      out[parts[0].strip()] = (int(parts[1]), float(parts[2]))
  return out


def main():
  fetch, write, steps, dst = read(sys.argv[1]), read(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
  kernels = {}
  for name in sorted(set(fetch) | set(write)):
    d = fetch.get(name, write.get(name))[0]
    kernels[name] = {'dispatches': d,
                     'fetch_bytes': fetch.get(name, (0, 0.0))[1] * 1024.0 * 2.0,
                     'write_bytes': write.get(name, (0, 0.0))[1] * 1024.0}
  json.dump({'steps_in_run': steps, 'fetch_correction': 'KiB x 2 (gfx950 128-B requests tallied as 64 B)',
             'write_correction': 'KiB x 1', 'kernels': kernels}, open(dst, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
  main()
