#!/bin/bash
# register / scratch / LDS usage of every kernel in a .hip file:  scratch/kres.sh speedplusbaseline_amd/csrc/x.hip [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re,subprocess
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur=subprocess.run(['/usr/bin/c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()
        cur=re.sub(r'\(anonymous namespace\)::','',cur); cur=re.sub(r'\(.*','',cur); vals={}
    l=l.split('remark:',1)[1] if 'remark:' in l else l
    for key in ('VGPRs','AGPRs','ScratchSize','Occupancy','TotalSGPRs','LDS Size','VGPRs Spill'):
        m=re.match(r'\s*'+key+r'( \[[^\]]*\])?: (\d+)',l)
        if m and cur: vals[key]=m.group(2)
    if 'LDS Size' in l and cur:
        print('%-70s vgpr %4s agpr %3s sgpr %3s scratch %4s occ %s lds %s'%(cur[:70],vals.get('VGPRs'),vals.get('AGPRs'),vals.get('TotalSGPRs'),vals.get('ScratchSize'),vals.get('Occupancy'),vals.get('LDS Size'))); cur=None
"
