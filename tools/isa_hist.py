"""Instruction histogram of one kernel in a hipcc -S --cuda-device-only listing.
usage: python tools/isa_hist.py engine.s <substring of mangled name> [top]"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
c = Counter()
for l in lines[start:end]:
    m = re.match(r"^\s+([a-z][a-z_0-9]+)\s", l)
    if m:
        c[m.group(1)] += 1
print(lines[start].split(":")[0], "instructions:", sum(c.values()))
for k, v in c.most_common(top):
    print("   %-30s %d" % (k, v))
