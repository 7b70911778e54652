#!/bin/bash
# tools/dbg/isa_dis.sh <unit> [kernel-symbol-substring]: disassemble the gfx950 code of /tmp/pobj/<unit>.o (objects of tools/dbg/build_fast.sh;
# units gpu tesa rd rd_lo), keep one kernel's listing in /tmp/dis/<unit>_k.s and print its instruction-class counts and register / spill
# figures.  Then: python tools/dbg/isa_waits.py /tmp/dis/<unit>_k.s  (loads in flight at every s_waitcnt vmcnt; sites with <= 2 in <file>.sites)
u=$1; k=${2:-_ZL17k_analyse_flow_rd}
mkdir -p /tmp/dis && cd /tmp/dis
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading /tmp/pobj/$u.o >/dev/null 2>&1
mv /tmp/pobj/$u.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 $u.co; rm -f /tmp/pobj/$u.o.0.host*
/opt/rocm/lib/llvm/bin/llvm-objdump -d $u.co > $u.s
a=$(grep -n "<$k" $u.s | head -1 | cut -d: -f1)
b=$(awk -v a=$a 'NR>a && /^[0-9a-f]+ <_Z/{print NR; exit}' $u.s)
[ -z "$b" ] && b=$(( $(wc -l < $u.s) + 1 ))          # the kernel is the last symbol of the code object
sed -n "${a},$((b-1))p" $u.s > ${u}_k.s
for p in global_load flat_load scratch_load scratch_store s_waitcnt v_readlane v_mul_lo_u32 s_cbranch "^\s*v_" "^\s*s_"; do echo "$p $(grep -c "$p" ${u}_k.s)"; done
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $u.co | grep "\.name:\|vgpr_count\|vgpr_spill\|sgpr_spill\|group_segment_fixed" | paste - - - - - | grep "$k" | sed 's/ \+/ /g'
