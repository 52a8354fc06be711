"""Summarise rocprofv3 --pmc passes (one directory per pass) for one kernel into profiles/.
usage: python tools/summarize_pmc.py gpurun_out/pmc_k2 "extprod_rotate_kernel<5, 5" profiles/r01_pmc_blind_rotate 16384
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
sys.path.insert(0, ".")
from tfhe_aes_amd import _build  # noqa: E402

out["engine_src_sha256"] = _build.engine_source_hash()       # bench.py attaches `traffic` only to runs of the same sources
json.dump(out, open(prefix + ".json", "w"), indent=1)
print(json.dumps(out, indent=1))
