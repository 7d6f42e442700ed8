#!/bin/bash
# tools/ab/asm_report.sh [extra flags]: compile gemm_f16x3.hip to /tmp/asm/gemm_v.s and print VGPR / scratch of the 3-term plane-output kernels
mkdir -p /tmp/asm
src=/root/repo/loco-asr_amd/csrc
flags=$(sed -n 's/^CXXFLAGS ?= //p' $src/Makefile | sed 's/$(ARCH)/gfx950/')
hipcc $flags "$@" -Rpass-analysis=kernel-resource-usage -S --cuda-device-only $src/gemm_f16x3.hip -o /tmp/asm/gemm_v.s 2> /tmp/asm/gemm_v.usage
grep "error" -A5 /tmp/asm/gemm_v.usage | head -20
python3 - <<'PY'
import re
t=open('/tmp/asm/gemm_v.usage').read()
seen=set()
for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", t, re.S):
    n=m.group(1)
    if 'gemm_f16x3_dma' in n:
        k=re.search(r'ILi(\d)ELb(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)',n)
        key=(k.group(3),k.group(4),k.group(5),k.group(6),k.group(9))
        if k.group(1) in '14' and key not in seen:
            seen.add(key); print('epi',k.group(1),'split',k.group(2),'WMxWN',k.group(3),k.group(4),'AST/WST',k.group(5),k.group(6),'NJ',k.group(7),'WPS',k.group(8),'terms',k.group(9),'| vgpr',m.group(2),'agpr',m.group(3),'scratch',m.group(4),'occ',m.group(5),'lds',m.group(6))
PY
L=$(grep -n "^_ZN4loco21gemm_f16x3_dma_kernelILi1ELb1ELi4ELi4ELi3ELi2ELi4ELi0ELi3ELb0EEEvNS_13GemmSplitArgsEiiiii:" /tmp/asm/gemm_v.s | cut -d: -f1)
awk -v s=$L 'NR>=s && NR<=s+6500' /tmp/asm/gemm_v.s > /tmp/asm/ffn1_v.s
grep -n "s_barrier\|^.LBB\|scratch_" /tmp/asm/ffn1_v.s | head -30
