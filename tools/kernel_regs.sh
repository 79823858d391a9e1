#!/bin/bash
# Register / LDS / scratch use of every kernel of a built library (reads the gfx950 code objects' metadata notes):
#   tools/kernel_regs.sh [path/to/libslideo_amd.so]
lib=$(readlink -f ${1:-$(dirname $0)/../slideo_amd/lib/libslideo_amd.so})
tmp=$(mktemp -d)
cd $tmp
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=fat.bin $lib
python3 - <<'PY'
import re,subprocess,sys
d=open('fat.bin','rb').read()
# clang offload bundles: find every ELF inside the fatbin
i=0;n=0
while True:
    i=d.find(b'\x7fELF',i)
    if i<0: break
    open('co%d.elf'%n,'wb').write(d[i:]); n+=1; i+=4
rows={}
import yaml
for k in range(n):
    out=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf','--notes','co%d.elf'%k],capture_output=True,text=True).stdout
    m=re.search(r'---\n(.*?)\n\.\.\.',out,re.S)
    if not m: continue
    y=yaml.safe_load(m.group(1))
    for c in y.get('amdhsa.kernels',[]): rows[c['.name']]=c
for name,c in sorted(rows.items(), key=lambda kv: kv[0]):
    dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip().split('(')[0]
    print('%-58s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s wg %s'%(dn[-58:],c.get('.vgpr_count'),c.get('.agpr_count',0),c.get('.sgpr_count'),c.get('.group_segment_fixed_size'),c.get('.private_segment_fixed_size'),c.get('.max_flat_workgroup_size')))
PY
rm -rf $tmp
