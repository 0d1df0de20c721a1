#!/bin/bash
# VGPR / spill summary of the kernels matching $1 (default: cg_fused) from the compiler's resource report of a fresh build (KRES_FLAGS: extra -D flags)
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
T="$(mktemp -d)"
make -s -C "$R/signed-heat-3d_amd/csrc" OUT="$T" HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-parameter -Wno-unused-function $KRES_FLAGS" || exit 1
python3 - "$T/kernel_resources.txt" "$1" <<'P'
import re,sys,subprocess
pat=sys.argv[2] if len(sys.argv)>2 and sys.argv[2] else "cg_fused"
cur=None
for line in open(sys.argv[1]):
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); d={}
    for k in ('VGPRs','AGPRs','VGPRs Spill','SGPRs','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(r'remark: [^ ]* +'+k+r': (\d+)',line)
        if m and cur: d[k]=m.group(1)
    if cur and 'LDS Size' in line and pat in cur:
        name=subprocess.run(['c++filt',cur],capture_output=True,text=True).stdout.strip()
        name=re.sub(r'\(.*','',name)
        print(name, ' '.join('%s=%s'%(k.split(' [')[0].replace('\\',''),v) for k,v in d.items()))
P
rm -rf "$T"
