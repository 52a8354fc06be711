"""Dynamic instruction count of one iteration of the K2 throughput kernel (R-ciphertext unit) from a hipcc -S listing.
Every basic block of the listing carries LLVM's loop annotation ("in Loop: Header=BBn_m Depth=d" / "Loop Header: Depth=d" /
"Parent Loop BBn_m Depth=d"): blocks of the unit's iteration loop count once, blocks of its rolled level loop (a child loop with a
large body) LEVELS-1 times.  Blocks of other small child loops and anything outside the iteration loop are not counted.
usage: python tools/k2_dyncount.py engine.s <substring of mangled name> [unit: 0 = first unit in the listing (R2), 1 = second (R)] [top]
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
LEVELS_M1 = 4

# pass 1: blocks = (first line, last line, header label of innermost loop or None, depth)
blocks = []
cur = None
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.\d+:", l)
    if m:
        if cur:
            cur[1] = i
            blocks.append(cur)
        ann = l
        j = i + 1
        while j < len(body) and body[j].lstrip().startswith(";") and not re.match(r"^; %bb", body[j]):
            ann += " " + body[j]
            j += 1
        hdr, depth, parent = None, 0, None
        mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", ann)
        if mm:
            hdr, depth = "." + "L" + mm.group(1), int(mm.group(2))
        mm = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", ann)
        if mm and m.group(1):
            hdr, depth = m.group(1), int(mm.group(1))
        mm = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", ann)
        if mm:
            parent = ".L" + mm.group(1)
        cur = [i, None, hdr, depth, parent]
if cur:
    cur[1] = len(body)
    blocks.append(cur)
# parent of every loop header
parent_of = {b[2]: b[4] for b in blocks if b[4] and b[2]}
size = Counter()
for b in blocks:
    if b[2]:
        size[b[2]] += b[1] - b[0]
# iteration loops: depth-1 loops with a large body (including children)
tot = Counter()
for h, n in size.items():
    tot[h] += n
    if h in parent_of:
        tot[parent_of[h]] += n
outer = [h for h in tot if h not in parent_of and tot[h] > 1500]
outer.sort(key=lambda h: next(b[0] for b in blocks if b[2] == h))
# round 6: the four-ciphertext body's iteration loop is no longer one annotated loop (its idle wavefronts run their own body behind a
# wave-uniform branch, and LLVM unswitches it): the six-ciphertext body is then the only large annotated loop -- take the last one
H = outer[min(unit, len(outer) - 1)]
level_loops = [h for h, p in parent_of.items() if p == H and size[h] > 300]


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


c, cyc = Counter(), Counter()
for b in blocks:
    if b[2] == H:
        w = 1
    elif b[2] in level_loops:
        w = LEVELS_M1
    else:
        continue
    for l in body[b[0]:b[1]]:
        m = re.match(r"^\s+([a-z][a-z_0-9]+)\s", l)
        if m:
            c[m.group(1)] += w
            cyc[m.group(1)] += w * cost(m.group(1))
n = sum(c.values())
valu = sum(v for k, v in c.items() if k.startswith("v_"))
f64 = sum(v for k, v in c.items() if "f64" in k)
lds = sum(v for k, v in c.items() if k.startswith("ds_"))
vmem = sum(v for k, v in c.items() if k.startswith(("buffer_", "global_")))
print("unit %d: per iteration: %d instructions, %d VALU (%d f64, %d other), %d LDS, %d VMEM, %d scalar/other" % (unit, n, valu, f64, valu - f64, lds, vmem, n - valu - lds - vmem))
print("estimated VALU issue cycles per wave per iteration: %d" % sum(cyc.values()))
for k, v in c.most_common(int(sys.argv[4]) if len(sys.argv) > 4 else 0):
    print("   %-28s %5d  %6d cyc" % (k, v, cyc[k]))
