#!/bin/bash
# VGPR / spill summary of the kernels matching $1 (default: cg_fused) -- hipcc -Rpass-analysis=kernel-resource-usage
cd "$(dirname "$0")/../signed-heat-3d_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $KRES_FLAGS -Rpass-analysis=kernel-resource-usage -shared shm_grid.hip -o /tmp/kres.so -ldl 2>/tmp/kres.txt
python3 - "$1" <<'P'
import re,sys
pat=sys.argv[1] if len(sys.argv)>1 and sys.argv[1] else "cg_fused"
cur=None
for line in open('/tmp/kres.txt'):
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); d={}
    for k in ('VGPRs','AGPRs','VGPRs Spill','SGPRs','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(r'remark: [^ ]* +'+k+r': (\d+)',line)
        if m and cur: d[k]=m.group(1)
    if cur and 'LDS Size' in line and pat in cur:
        import subprocess
        name=subprocess.run(['c++filt',cur],capture_output=True,text=True).stdout.strip()
        name=re.sub(r'\(.*','',name)
        print(name, ' '.join('%s=%s'%(k.split(' [')[0].replace('\\',''),v) for k,v in d.items()))
P
