"""Dynamic instruction count of one iteration of the K2 throughput kernel (R-ciphertext unit) from a hipcc -S listing.
The kernel body is: outer loop over the 669 iterations (label A), inside it one rolled loop over LEVELS-1 levels (label B).
usage: python tools/k2_dyncount.py engine.s <substring of mangled name> [unit index: 0 = first unit in the listing (R2), 1 = second (R)]
Issue cost per wave-instruction (cycles) from tools/ubench/ubench_ops (gfx950): f64 4; shifts/bfe/64-bit/add_co 4; simple 32-bit 2."""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
unit = int(sys.argv[3]) if len(sys.argv) > 3 else 1
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start, len(src)) if ".amdhsa_kernel" in src[i])
body = src[start:end]
# the iteration loop of a unit = a loop header (any depth) whose body is large and which has a rolled child loop (the levels);
# units in listing order (the R2 unit first, then the R unit)
def _label_line(i):           # the ".LBBn_m:" line a "Loop Header" comment belongs to (the comment may sit on a continuation line)
    while not body[i].startswith(".LBB"):
        i -= 1
    return i


hdr = [(_label_line(i), int(re.search(r"Loop Header: Depth=(\d+)", l).group(1))) for i, l in enumerate(body) if re.search(r"Loop Header: Depth=\d+", l)]
cands = []
for n, (i, d) in enumerate(hdr):
    lab = body[i].split(":")[0].strip()
    end = next((j for j in range(i + 1, len(body)) if re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", body[j])), None)
    if end is None or end - i < 1500:
        continue
    kids = [(j, dd) for j, dd in hdr if i < j < end and dd == d + 1]
    big = [(j, dd) for j, dd in kids if next((e for e in range(j + 1, end) if re.search(r"s_c?branch\w*\s+" + re.escape(body[j].split(":")[0].strip()) + r"\b", body[e])), j) - j > 300]
    if big:
        cands.append((i, end, big[0][0]))
h, h_end, b = cands[unit]
lab_b = body[b].split(":")[0].strip()
b_end = next(i for i in range(b, len(body)) if re.search(r"s_cbranch_\w+\s+" + re.escape(lab_b) + r"\b", body[i]))
LEVELS_M1 = 4


def cost(op):
    if op.startswith(("s_", "ds_", "buffer_", "global_", "scratch_")):
        return 0
    if "f64" in op:
        return 4
    if op in ("v_add_u32_e32", "v_sub_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_mov_b32_e32", "v_cndmask_b32_e32", "v_cndmask_b32_e64",
              "v_subrev_u32_e32", "v_not_b32_e32", "v_add_u32_e64", "v_sub_u32_e64", "v_cmp_eq_u32_e32", "v_cmp_eq_u32_e64", "v_cmp_ne_u32_e32",
              "v_cmp_lt_u32_e32", "v_cmp_gt_u32_e32", "v_and_or_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_bitop3_b32", "v_add3_u32", "v_or3_b32", "v_xad_u32"):
        return 2
    return 4


def count(lo, hi, mult, c, cyc):
    for l in body[lo:hi]:
        m = re.match(r"^\s+([a-z][a-z_0-9]+)\s", l)
        if m:
            c[m.group(1)] += mult
            cyc[m.group(1)] += mult * cost(m.group(1))


c, cyc = Counter(), Counter()
count(h, b, 1, c, cyc)
count(b, b_end + 1, LEVELS_M1, c, cyc)
count(b_end + 1, h_end + 1, 1, c, cyc)
tot = sum(c.values())
valu = sum(v for k, v in c.items() if k.startswith("v_"))
f64 = sum(v for k, v in c.items() if "f64" in k)
lds = sum(v for k, v in c.items() if k.startswith("ds_"))
vmem = sum(v for k, v in c.items() if k.startswith(("buffer_", "global_")))
print("unit %d: per iteration: %d instructions, %d VALU (%d f64, %d other), %d LDS, %d VMEM, %d scalar/other" % (unit, tot, valu, f64, valu - f64, lds, vmem, tot - valu - lds - vmem))
print("estimated VALU issue cycles per wave per iteration: %d" % sum(cyc.values()))
for k, v in c.most_common(int(sys.argv[4]) if len(sys.argv) > 4 else 45):
    print("   %-28s %5d  %6d cyc" % (k, v, cyc[k]))
