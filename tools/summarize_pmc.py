"""Summarise rocprofv3 --pmc passes (one directory per pass) for one kernel into profiles/.
usage: python tools/summarize_pmc.py gpurun_out/pmc_k2 "blind_rotate16_kernel<5, 5" profiles/r03_pmc_blind_rotate 16384 [CUs [valu_cycles_per_wave_iteration flops_per_ciphertext_iteration]]
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KB and reports half of a wide coalesced
read stream on gfx950 (x2), WRITE_SIZE is exact for 16-byte stores; separate --pmc passes."""
import csv
import glob
import json
import sys
from collections import defaultdict

src, key, prefix, bits = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
tot, n = defaultdict(float), defaultdict(int)
dur = []
for f in glob.glob(src + "/*/*/*_counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add(r["Dispatch_Id"])
    for f2 in glob.glob(f.rsplit("/", 1)[0] + "/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f2)):
            if key in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for c in list(tot):
        n[c] = max(n[c], len(seen))
launch_ms = sum(dur) / len(dur)
per = {c: v / max(1, n[c]) for c, v in tot.items()}
fetch = per.get("FETCH_SIZE", 0) * 1024 * 2
write = per.get("WRITE_SIZE", 0) * 1024
out = {
    "kernel": key, "params": "PARAM_OPT", "bits_per_launch": bits, "avg_launch_ms_profiled": launch_ms,
    "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
    "counters_per_launch": per,
}
if "GRBM_GUI_ACTIVE" in per:
    out["effective_clock_ghz"] = per["GRBM_GUI_ACTIVE"] / 8 / (launch_ms * 1e-3) / 1e9
if "TCC_HIT_sum" in per:
    out["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
if "SQ_WAVE_CYCLES" in per:
    w = per["SQ_WAVE_CYCLES"]
    out["wave_time_split"] = {"active": per["SQ_ACTIVE_INST_ANY"] / w, "wait_inst": per["SQ_WAIT_INST_ANY"] / w, "wait_any": per["SQ_WAIT_ANY"] / w}
if "SQ_ACTIVE_INST_VALU" in per and "GRBM_GUI_ACTIVE" in per:
    # SQ_ACTIVE_INST_* count in units of 4 cycles per SIMD; GRBM_GUI_ACTIVE sums the 8 XCDs' clocks; 4 SIMDs x CUs
    cus = float(sys.argv[5]) if len(sys.argv) > 5 else 256.0
    out["valu_busy"] = per["SQ_ACTIVE_INST_VALU"] * 4.0 / (per["GRBM_GUI_ACTIVE"] / 8.0 * cus * 4.0)
    if "SQ_ACTIVE_INST_LDS" in per:
        out["lds_issue_share_of_wave_time"] = per.get("SQ_WAIT_INST_LDS", 0.0) / per["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in per else None
if len(sys.argv) > 6:
    # ceiling: algorithmic flops per workgroup-iteration / (VALU issue cycles of its 4 waves x 32 flop per SIMD cycle);
    # argv[6] = estimated VALU issue cycles per wave per iteration (tools/k2_dyncount.py), argv[7] = algorithmic flops per ciphertext-iteration
    cyc, fl = float(sys.argv[6]), float(sys.argv[7])
    out["valu_issue_cycles_per_wave_iteration"] = cyc
    out["ceiling_frac"] = 3.0 * fl / (4.0 * cyc * 32.0)
sys.path.insert(0, ".")
from tfhe_aes_amd import _build  # noqa: E402

out["engine_src_sha256"] = _build.engine_source_hash()       # bench.py attaches `traffic` only to runs of the same sources
json.dump(out, open(prefix + ".json", "w"), indent=1)
print(json.dumps(out, indent=1))
