"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stderr log) per kernel."""
import re
import sys

log = open(sys.argv[1]).read()
blocks = re.split(r'remark: Function Name: ', log)[1:]
keys = [("VGPR", r'    VGPRs'), ("AGPR", r'AGPRs'), ("spill", r'VGPRs Spill'), ("scratch", r'ScratchSize \[bytes/lane\]'),
        ("occ", r'Occupancy \[waves/SIMD\]'), ("LDS", r'LDS Size \[bytes/block\]'), ("SGPR", r'TotalSGPRs')]
for b in blocks:
    name = b.split()[0]
    vals = []
    for label, k in keys:
        m = re.search(k + r': (\d+)', b)
        vals.append("%s %s" % (label, m.group(1) if m else "?"))
    print("%-62s %s" % (name[:62], "  ".join(vals)))
