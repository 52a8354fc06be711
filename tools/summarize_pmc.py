"""Summarise rocprofv3 --pmc passes (one directory per pass) for one kernel into profiles/.
usage: python tools/summarize_pmc.py <pmc dir> "<kernel name substring>" profiles/<name> <bits per launch>
           [--cus 256] [--dyncount <file written by tools/k2_dyncount.py>] [--flops-per-ct-iteration 609280] [--cts-per-wg 6] [--waves-per-wg 8]
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KB and reports half of a wide coalesced read stream on gfx950
(x2), WRITE_SIZE is exact for 16-byte stores; separate --pmc passes.

Derived fields (round 4: each one is what its name says):
  effective_clock_ghz          GRBM_GUI_ACTIVE / 8 XCDs / launch time
  valu_issue_occupancy_model   share of SIMD cycles with a vector instruction issuing, MODELLED: SQ_INSTS_VALU (measured, per wave) x the
                               average issue cost per vector instruction of this kernel's loop (tools/k2_dyncount.py: ISA count x
                               per-opcode cycles of tools/ubench/ubench_ops) / (CUs x 4 SIMDs x cycles).  gfx950 has no busy-cycle
                               counter for the vector ALU that rocprofv3 lists: SQ_ACTIVE_INST_VALU returns the instruction count
                               (it equals SQ_INSTS_VALU to the digit), so round 3's `valu_busy` (that count x an assumed 4 cycles)
                               overstated the occupancy.
  ceiling_frac                 fraction of the f64 vector roof this instruction stream reaches with a vector instruction issuing on
                               every SIMD cycle at 2.4 GHz: (ciphertexts per workgroup x algorithmic flops per ciphertext-iteration) /
                               (wavefronts per workgroup x issue cycles per wave-iteration x 32 flop per SIMD-cycle)
  model_frac                   ceiling_frac x valu_issue_occupancy_model x clock / 2.4: what the three say the roofline fraction is
  l1_fill_bytes_per_clk_per_cu TCP_TCC_READ_REQ_sum x 128 B / CUs / cycles (the vector-memory path into a CU; ~50 is its ceiling)
"""
import argparse
import csv
import glob
import json
import re
import sys
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("src"); ap.add_argument("key"); ap.add_argument("prefix"); ap.add_argument("bits", type=float)
ap.add_argument("--cus", type=float, default=256.0)
ap.add_argument("--dyncount", default=None)
ap.add_argument("--flops-per-ct-iteration", type=float, default=609280.0)
ap.add_argument("--cts-per-wg", type=float, default=6.0)
ap.add_argument("--waves-per-wg", type=float, default=8.0)
ap.add_argument("--algorithmic-bytes", type=float, default=None)
args = ap.parse_args()

tot, n = defaultdict(float), defaultdict(int)
dur = []
for f in glob.glob(args.src + "/*/*/*_counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        if args.key in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add(r["Dispatch_Id"])
    for f2 in glob.glob(f.rsplit("/", 1)[0] + "/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f2)):
            if args.key in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for c in list(tot):
        n[c] = max(n[c], len(seen))
if not dur:
    sys.exit("no dispatch of a kernel matching %r under %s" % (args.key, args.src))
launch_ms = sum(dur) / len(dur)
per = {c: v / max(1, n[c]) for c, v in tot.items()}
fetch = per.get("FETCH_SIZE", 0) * 1024 * 2
write = per.get("WRITE_SIZE", 0) * 1024
out = {
    "kernel": args.key, "params": "PARAM_OPT", "bits_per_launch": args.bits, "avg_launch_ms_profiled": launch_ms,
    "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
    "counters_per_launch": per,
}
if args.algorithmic_bytes:
    out["algorithmic_bytes_per_launch"] = args.algorithmic_bytes
    out["traffic_over_algorithmic"] = (fetch + write) / args.algorithmic_bytes
cycles = None
if "GRBM_GUI_ACTIVE" in per:
    cycles = per["GRBM_GUI_ACTIVE"] / 8.0                      # the counter sums the 8 XCDs' clocks
    out["effective_clock_ghz"] = cycles / (launch_ms * 1e-3) / 1e9
if "TCC_HIT_sum" in per:
    out["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
if "SQ_WAVE_CYCLES" in per:
    w = per["SQ_WAVE_CYCLES"]
    out["wave_time_split"] = {"active": per["SQ_ACTIVE_INST_ANY"] / w, "wait_inst": per["SQ_WAIT_INST_ANY"] / w, "wait_any": per["SQ_WAIT_ANY"] / w}
    if "SQ_WAIT_INST_LDS" in per:
        out["lds_issue_share_of_wave_time"] = per["SQ_WAIT_INST_LDS"] / w
if "TCP_TCC_READ_REQ_sum" in per and cycles:
    out["l1_fill_bytes_per_clk_per_cu"] = per["TCP_TCC_READ_REQ_sum"] * 128.0 / args.cus / cycles
if args.dyncount:
    text = open(args.dyncount).read()
    m = re.search(r"per iteration: (\d+) instructions, (\d+) VALU", text)
    c = re.search(r"estimated VALU issue cycles per wave per iteration: (\d+)", text)
    n_valu, cyc = float(m.group(2)), float(c.group(1))
    out["valu_issue_cycles_per_wave_iteration"] = cyc
    out["valu_instructions_per_wave_iteration"] = n_valu
    out["ceiling_frac"] = args.cts_per_wg * args.flops_per_ct_iteration / (args.waves_per_wg * cyc * 32.0)
    if "SQ_INSTS_VALU" in per and cycles:
        out["sq_insts_valu"] = per["SQ_INSTS_VALU"]
        out["valu_issue_occupancy_model"] = per["SQ_INSTS_VALU"] * (cyc / n_valu) / (args.cus * 4.0 * cycles)
        out["model_frac"] = out["ceiling_frac"] * out["valu_issue_occupancy_model"] * out["effective_clock_ghz"] / 2.4
sys.path.insert(0, ".")
from tfhe_aes_amd import _build  # noqa: E402

out["engine_src_sha256"] = _build.engine_source_hash()       # bench.py attaches `traffic` only to runs of the same sources
json.dump(out, open(args.prefix + ".json", "w"), indent=1)
print(json.dumps(out, indent=1))
