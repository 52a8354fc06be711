"""Summarise the rocprofv3 --pmc passes of tools/pmc_k3.sh for the packing key switch (K3) into profiles/<name>.json.
usage: python tools/summarize_pmc_k3.py <pmc dir> profiles/<name> <bits per launch>
HBM bytes as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KB) x 1024 x 2 + WRITE_SIZE (KB) x 1024, separate passes.
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (CUs x 4 SIMDs x shader cycles of the launch) (the counter counts cycles, not quad-cycles)."""
import csv
import glob
import json
import sys
from collections import defaultdict

src, prefix, bits = sys.argv[1], sys.argv[2], float(sys.argv[3])
KEY = "keyswitch_mfma_lds_kernel<2>"
tot, n, dur = defaultdict(float), defaultdict(int), []
for f in glob.glob(src + "/*/*/*_counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        if KEY in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add(r["Dispatch_Id"])
    for f2 in glob.glob(f.rsplit("/", 1)[0] + "/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f2)):
            if KEY in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for c in list(tot):
        n[c] = max(n[c], len(seen))
if not dur:
    sys.exit("no dispatch of %s under %s" % (KEY, src))
launch_ms = sum(dur) / len(dur)
per = {c: v / max(1, n[c]) for c, v in tot.items()}
fetch, write = per.get("FETCH_SIZE", 0) * 1024 * 2, per.get("WRITE_SIZE", 0) * 1024
out = {"kernel": KEY, "params": "PARAM_OPT", "bits_per_launch": bits, "avg_launch_ms_profiled": launch_ms, "hbm_bytes_per_launch": fetch + write,
       "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "counters_per_launch": per}
cycles = None
if "GRBM_GUI_ACTIVE" in per:
    cycles = per["GRBM_GUI_ACTIVE"] / 8.0
    out["effective_clock_ghz"] = cycles / (launch_ms * 1e-3) / 1e9
if "TCC_HIT_sum" in per:
    out["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
if "SQ_WAVE_CYCLES" in per:
    w = per["SQ_WAVE_CYCLES"]
    out["wave_time_split"] = {"active": per["SQ_ACTIVE_INST_ANY"] / w, "wait_inst": per["SQ_WAIT_INST_ANY"] / w, "wait_any": per["SQ_WAIT_ANY"] / w}
    if per.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_frac"] = per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"]
if "SQ_VALU_MFMA_BUSY_CYCLES" in per and cycles:
    out["mfma_busy_frac"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (256.0 * 4.0 * cycles)
sys.path.insert(0, ".")
from tfhe_aes_amd import _build  # noqa: E402

out["engine_src_sha256"] = _build.engine_source_hash()
json.dump(out, open(prefix + ".json", "w"), indent=1)
print(json.dumps(out, indent=1))
