#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin or file): one line per kernel."""
import re, sys
t = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
OCC, LDS, SCR = r'Occupancy \[waves/SIMD\]', r'LDS Size \[bytes/block\]', r'ScratchSize \[bytes/lane\]'
for b in re.split(r'remark: [^\n]*Function Name: ', t)[1:]:
    name = b.split(' ')[0]
    if pat not in name:
        continue
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    print("VGPR %3s AGPR %3s SGPR %3s occ %s LDS %6s scratch %s  %s" % (g('VGPRs'), g('AGPRs'), g('SGPRs'), g(OCC), g(LDS), g(SCR), name))
