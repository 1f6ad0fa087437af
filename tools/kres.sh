#!/bin/bash
# registers / LDS / spills of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage): tools/kres.sh [pattern] [-DFLAG ...]
cd "$(dirname "$0")/.."
pat=${1:-k_}; shift
{ hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-honor-nans "$@" -c -o /tmp/kres.o transport_se_amd/csrc/tse_api.hip -Rpass-analysis=kernel-resource-usage 2>&1;
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-honor-nans -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c -o /tmp/kres3.o transport_se_amd/csrc/tse_stage3.hip -Rpass-analysis=kernel-resource-usage 2>&1; } | \
  python3 -c "
import re,sys
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?)\s+\[-Rpass', l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'): cur=t.split(':',1)[1].strip(); rows[cur]={}
    elif cur and ':' in t: k,v=t.split(':',1); rows[cur][k.strip()]=v.strip()
import subprocess
for n,r in rows.items():
    d=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
    d=re.sub(r'\(.*','',d).replace('void tse::','')
    if '$pat' in d: print('%-46s VGPR %-4s AGPR %-3s spill %-3s scratch %-4s LDS %-6s occ %s'%(d,r.get('VGPRs'),r.get('AGPRs'),r.get('VGPRs Spill'),r.get('ScratchSize [bytes/lane]'),r.get('LDS Size [bytes/block]'),r.get('Occupancy [waves/SIMD]')))
"
