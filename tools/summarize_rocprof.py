"""Copy a rocprofv3 --kernel-trace --stats run into profiles/ with a readable per-kernel summary.

usage: python tools/summarize_rocprof.py gpurun_out/prof_r01/<host> profiles/r01_v1 "comment"
Writes <prefix>_kernel_stats.csv (verbatim), <prefix>_summary.md (stats + the dispatches of the timed step,
i.e. the largest-grid launch of every kernel).
"""
import csv
import glob
import shutil
import sys
from collections import defaultdict

src, prefix = sys.argv[1], sys.argv[2]
comment = sys.argv[3] if len(sys.argv) > 3 else ""
stats = glob.glob(src + "/*_kernel_stats.csv")[0]
trace = glob.glob(src + "/*_kernel_trace.csv")[0]
shutil.copy(stats, prefix + "_kernel_stats.csv")
rows = list(csv.DictReader(open(trace)))
by = defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append(r)
with open(prefix + "_summary.md", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary\n\n%s\n\n" % comment)
    f.write("## kernel stats (whole process: key upload + key expansion + timed step)\n\n")
    f.write("| kernel | calls | total ms | avg ms | max ms | % |\n|---|---|---|---|---|---|\n")
    for r in csv.DictReader(open(stats)):
        f.write("| `%s` | %s | %.3f | %.3f | %.3f | %s |\n" % (r["Name"][:70], r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                          float(r["AverageNs"]) / 1e6, int(r["MaxNs"]) / 1e6, r["Percentage"]))
    f.write("\n## launches of the timed step (largest grid of each kernel)\n\n")
    f.write("| kernel | launches | grid | wg | VGPR | scratch B/lane | LDS B | avg ms | min ms | max ms |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for name, rs in sorted(by.items()):
        gmax = max(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) for r in rs)
        big = [r for r in rs if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) == gmax]
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in big]
        r0 = big[0]
        f.write("| `%s` | %d | %sx%sx%s | %s | %s | %s | %s | %.3f | %.3f | %.3f |\n" % (
            name[:60], len(big), r0["Grid_Size_X"], r0["Grid_Size_Y"], r0["Grid_Size_Z"], r0["Workgroup_Size_X"],
            r0["VGPR_Count"], r0["Scratch_Size"], r0["LDS_Block_Size"], sum(d) / len(d), min(d), max(d)))
print(open(prefix + "_summary.md").read())
