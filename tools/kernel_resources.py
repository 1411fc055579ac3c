#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage), cross-compiled
for gfx950 without a GPU.  usage: python tools/kernel_resources.py mix_stage_amd/csrc/conv16.hip [name-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
  src = sys.argv[1]
  flt = sys.argv[2] if len(sys.argv) > 2 else ''
  cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '--cuda-device-only',
         '-I' + os.path.join(ROOT, 'include'), '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null']
  out = subprocess.run(cmd, capture_output=True, text=True).stderr
  cur = None
  rows = []
  for line in out.splitlines():
    m = re.search(r'remark: (?:\s*)Function Name: (\S+)', line)
    if m:
      cur = {'name': subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()}
      rows.append(cur)
      continue
    m = re.search(r'remark:\s+(\w[\w ]*\w)(?: \[bytes/(?:lane|workgroup)\])?: (\d+)', line)
    if m and cur is not None:
      cur[m.group(1)] = int(m.group(2))
  print('%-90s %5s %5s %5s %6s %6s %4s %7s' % ('kernel', 'SGPR', 'VGPR', 'AGPR', 'scratch', 'spillV', 'occ', 'LDS'))
  for r in rows:
    if flt and flt not in r['name']:
      continue
    nm = re.sub(r'^void ms::', '', r['name'])
    nm = re.sub(r'\(.*$', '', nm)
    print('%-90s %5d %5d %5d %6d %6d %4d %7d' % (nm[:90], r.get('TotalSGPRs', -1), r.get('VGPRs', -1), r.get('AGPRs', -1),
                                                r.get('ScratchSize', -1), r.get('VGPR Spill', -1), r.get('Occupancy', -1),
                                                r.get('LDS Size', -1)))


if __name__ == '__main__':
  main()
