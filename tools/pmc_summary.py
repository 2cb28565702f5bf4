"""Summarise rocprofv3 passes of one command into a JSON for profiles/: per-launch means of every counter for the dominant
kernel (regex), kernel time from the --stats pass, HBM bytes per launch with the gfx950 correction the guide prescribes
(MI355X_MICROARCH.md, HBM: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> x2; WRITE_SIZE exact; both
in KB).
    python tools/pmc_summary.py <dir with stats/ pmc1/ pmc2/ pmc3/> <kernel regex> <out.json> [note] [skip]
`skip`: leave out the first `skip` launches of the kernel in every pass (a first launch that does other work, e.g. a walk
without history).
"""
import csv, glob, json, os, re, sys

root, rx, out = sys.argv[1], re.compile(sys.argv[2]), sys.argv[3]
note = sys.argv[4] if len(sys.argv) > 4 else ""
skip = int(sys.argv[5]) if len(sys.argv) > 5 else 0
res = {"kernel_regex": sys.argv[2], "note": note, "source": "rocprofv3 (ROCm 7.2), MI355X; passes: --kernel-trace --stats | --pmc FETCH_SIZE | "
       "--pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum | --pmc GRBM_GUI_ACTIVE SQ_*; every --pmc pass with --kernel-trace only"}


def rows(pattern):
    for f in glob.glob(os.path.join(root, pattern), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)


# kernel time from the trace (no counters attached)
durs, name = [], None
for r in sorted(rows("stats/**/*kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"])):
    if rx.search(r["Kernel_Name"]):
        durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        name = r["Kernel_Name"]
durs = durs[skip:]
if skip:
    res["launches_skipped"] = skip
if durs:
    res["kernel"] = name[:200]
    res["launches"] = len(durs)
    res["kernel_ms_avg"] = sum(durs) / len(durs)
    res["kernel_ms_min"] = min(durs)
    res["kernel_ms_max"] = max(durs)
counters = {}
for p in ("pmc1", "pmc2", "pmc3"):
    acc = {}
    for r in sorted(rows(p + "/**/*counter_collection.csv"), key=lambda r: int(r["Dispatch_Id"])):
        if rx.search(r["Kernel_Name"]):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[skip:]
        if not v:
            continue
        counters[k] = sum(v) / len(v)
        counters[k + "_launches"] = len(v)
res["counters_per_launch"] = counters
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    res["FETCH_SIZE_KB"] = counters["FETCH_SIZE"]
    res["WRITE_SIZE_KB"] = counters["WRITE_SIZE"]
    res["hbm_bytes_per_launch_uncorrected"] = (counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024
    res["hbm_bytes_per_launch"] = (2 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024
    res["traffic_note"] = ("(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B: the guide's gfx950 correction (FETCH_SIZE tallies 128-B requests at 64 B) "
                           "is calibrated for 16 B/lane streaming reads; other widths are uncalibrated, so the read side lies between the "
                           "uncorrected and the corrected figure")
if "TCC_HIT_sum" in counters and "TCC_MISS_sum" in counters and counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"] > 0:
    res["l2_hit_rate"] = counters["TCC_HIT_sum"] / (counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"])
if "GRBM_GUI_ACTIVE" in counters and durs:
    res["effective_clock_GHz"] = counters["GRBM_GUI_ACTIVE"] / 8 / (res["kernel_ms_avg"] * 1e-3) / 1e9
if "SQ_ACTIVE_INST_VALU" in counters and "GRBM_GUI_ACTIVE" in counters:
    # SQ_* cycle counters are in quad-cycles summed over SIMDs; 256 CUs x 4 SIMDs, GRBM summed over 8 XCDs
    simd_cycles = counters["GRBM_GUI_ACTIVE"] / 8 * 1024
    res["valu_busy_fraction_of_simd_cycles"] = counters["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in res if k != "counters_per_launch"}))
