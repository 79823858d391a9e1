#!/bin/bash
# ISA of one kernel of a built stage object:  tools/kernel_isa.sh <stage object, e.g. stage_knn> <kernel name substring> [lib obj dir]
obj=$(readlink -f ${3:-$(dirname $0)/../slideo_amd/lib/obj}/$1.o)
tmp=$(mktemp -d); cd $tmp
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=fat.bin $obj
python3 -c "
d=open('fat.bin','rb').read(); i=d.find(b'\x7fELF'); open('co.elf','wb').write(d[i:])"
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 co.elf 2>/dev/null | sed 's/\/\/.*//' > all.s
S=$(grep -n "^[0-9a-f]* <.*$2" all.s | head -1 | cut -d: -f1)
E=$(awk -v s=$S 'NR>s && /^[0-9a-f]+ </{print NR; exit}' all.s)
sed -n "${S},${E:-\$}p" all.s
rm -rf $tmp
