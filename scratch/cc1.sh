#!/bin/bash
# scratch: compile ONE csrc file with resource-usage remarks and print every kernel's registers / scratch (usage: cc1.sh hugs_x.hip [filter])
cd /root/repo/nerf-hugs_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage $EXTRA -c $1 -o /tmp/cc1.o 2> /tmp/cc1.res
grep -v remark /tmp/cc1.res | grep -v "^ \|\^" | head -20
python3 - "$2" <<'PY'
import re,subprocess,sys
txt=open('/tmp/cc1.res').read()
rec=[];cur=None
for m in re.finditer(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", txt):
    k,v=m.groups()
    if k=='Function Name': cur=[v];rec.append(cur)
    else: cur.append(f"{k.split()[0]}={v}")
names=subprocess.run(['c++filt']+[r[0] for r in rec],capture_output=True,text=True).stdout.split('\n')
for r,n in zip(rec,names):
    n=re.sub(r"\(.*","",n)
    if not sys.argv[1] or sys.argv[1] in n: print(n,' '.join(r[1:]))
PY
