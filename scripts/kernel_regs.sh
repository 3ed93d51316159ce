#!/bin/bash
# register / scratch / LDS / occupancy figures of the kernels of one source file (compiler remarks), filtered by a pattern:
#   scripts/kernel_regs.sh conv gemm3_kernel
f=$1; pat=${2:-.}; shift; shift
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -Rpass-analysis=kernel-resource-usage -c -o /dev/null srl_amd/csrc/$f.hip 2>&1 | \
  python3 -c "
import sys,re
cur={}
for line in sys.stdin:
    m=re.search(r'remark: (.*)',line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        cur={'name':t.split(':',1)[1].strip()}
    else:
        kv=t.split(':')
        if len(kv)==2: cur[kv[0].strip()]=kv[1].strip()
        if kv[0].strip().startswith('LDS Size'):
            print(cur.get('name','?')[:60], '| vgpr',cur.get('VGPRs'),'agpr',cur.get('AGPRs'),'scratch',cur.get('ScratchSize [bytes/lane]'),'occ',cur.get('Occupancy [waves/SIMD]'),'lds',cur.get('LDS Size [bytes/block]'))
" | grep -E "$pat"
