#!/bin/bash
# Per-kernel register / scratch / LDS table of the library (clang's -Rpass-analysis=kernel-resource-usage over every .hip source,
# with the Makefile's flags) -> stdout.  Round 3 found 3 % of step time in this table (spills into K-loops); re-take it after every
# change to a shared epilogue.     usage: bash tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
cd "$(dirname "$0")/../iccv2025-gdl_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -Wno-unused-function"
printf "%-22s %-118s %6s %6s %8s %8s %5s\n" source kernel VGPRs AGPRs scratchB LDS_B occ
for f in *.hip; do
  hipcc $F -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys, subprocess
cur = {}
rows = []
for l in sys.stdin:
    m = re.search(r'remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)', l)
    if not m: continue
    k, v = m.groups()
    if k == 'Function Name':
        cur = {'name': v}
        rows.append(cur)
    else:
        cur[k.split(' ')[0]] = v
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace('gdl::', '').split('(')[0]
    print('%-22s %-118s %6s %6s %8s %8s %5s' % ('$f', n[:118], r.get('VGPRs'), r.get('AGPRs'), r.get('ScratchSize'), r.get('LDS'), r.get('Occupancy')))
"
done
