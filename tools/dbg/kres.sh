#!/bin/bash
# VGPRs / occupancy / spills of the k_uscore instantiations of the working tree (cross-compiles; no GPU needed).
R=$(cd $(dirname $0)/../.. && pwd)
cd $R/nextsearch-api_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed ${EXTRA:-} -I../include -Ihost -Icsrc -shared -o /tmp/kres.so csrc/ns_api.hip -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for ln in sys.stdin:
    m=re.search(r'remark:\s+(.*?)\s*\[-Rpass', ln)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name:'):
        cur=t.split(':',1)[1].strip(); rows[cur]={}
    elif cur and ':' in t:
        k,v=t.split(':',1); rows[cur][k.strip()]=v.strip()
pat=sys.argv[1] if len(sys.argv)>1 else 'k_uscore'
for n,r in rows.items():
    if pat in n:
        short=re.sub(r'EEvPK.*','',n).replace('_ZN2ns8','')
        print(f\"{short:60s} VGPR {r.get('VGPRs'):>4} occ {r.get('Occupancy [waves/SIMD]'):>2} scratch {r.get('ScratchSize [bytes/lane]'):>4} sgpr_spill {r.get('SGPRs Spill'):>3} vgpr_spill {r.get('VGPRs Spill'):>3} lds {r.get('LDS Size [bytes/block]')}\")
" "${1:-k_uscore}"
