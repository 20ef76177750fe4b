"""Condenses the rocprofv3 output of scripts/collect_profiles.sh into the small files kept under profiles/."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "den_tied_kernel"  # the dominant kernel of the default bench workload


def find(sub, pattern):
    hits = glob.glob(os.path.join(out, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


summary = {}
# ---- kernel stats: keep the rows of our kernels only (torch's RNG kernel names are pages long)
f = find("kt", "*kernel_stats.csv")
if f:
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if r and ("tc::" in r[0])]
    with open(os.path.join(out, "%s_kernel_stats.csv" % tag), "w", newline="") as g:
        csv.writer(g, quoting=csv.QUOTE_ALL).writerows(keep)
    for r in keep[1:]:
        if KERNEL in r[0]:
            summary["kernel"] = r[0]
            summary["calls"] = int(r[1])
            summary["avg_ns"] = float(r[3])
            summary["min_ns"] = float(r[5])


f = find("kt", "*kernel_trace.csv")
if f:
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
               for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"])
    dur = [x[1] for x in d]
    if dur:
        srt = sorted(dur)
        summary["trace_median_ns"] = srt[len(srt) // 2]
        summary["trace_mean_last20_ns"] = sum(dur[-20:]) / len(dur[-20:])  # the launches bench.py times with HIP events
        summary["trace_first8_ns"] = dur[:8]


def counter_means(sub):
    f = find(sub, "*counter_collection.csv")
    res = {}
    if not f:
        return res
    rd = csv.DictReader(open(f))
    acc = {}
    for row in rd:
        if KERNEL not in row.get("Kernel_Name", ""):
            continue
        name, val = row["Counter_Name"], float(row["Counter_Value"])
        key = (name, row.get("Dispatch_Id"))
        acc[key] = acc.get(key, 0.0) + val  # sum over dimensions (XCDs / SEs) of one dispatch
    per = {}
    for (name, _), v in acc.items():
        per.setdefault(name, []).append(v)
    for name, vs in per.items():
        res[name] = sum(vs) / len(vs)
    return res


fetch, write = counter_means("fetch"), counter_means("write")
if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
    summary["fetch_size_kb_per_launch"] = fetch["FETCH_SIZE"]
    summary["write_size_kb_per_launch"] = write["WRITE_SIZE"]
    summary["hbm_bytes_per_launch"] = (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0
sq = {}
for sub in ("sq1", "sq2", "sq3"):
    sq.update(counter_means(sub))
if not sq and os.environ.get("SQ_CSV"):
    # the counters of an earlier collection (a clock-only re-run: CLOCK_ONLY=1 scripts/collect_profiles.sh)
    for row in csv.DictReader(open(os.environ["SQ_CSV"])):
        sq[row["counter"]] = float(row["mean_per_launch"])
    if os.environ.get("KERNEL_MS"):
        summary["avg_ns"] = float(os.environ["KERNEL_MS"]) * 1e6
elif sq:
    with open(os.path.join(out, "%s_sq_counters.csv" % tag), "w") as g:
        g.write("counter,mean_per_launch\n")
        for k in sorted(sq):
            g.write("%s,%.6g\n" % (k, sq[k]))
# ---- secondary ceilings: what the kernel's LDS and VALU pipes were busy for, as time at the measured clock
config = sys.argv[4] if len(sys.argv) > 4 else "C3"
clock_mhz = None
cf = os.path.join(out, "clock.txt")
if os.path.exists(cf):
    import re
    m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", open(cf).read())
    if m:
        clock_mhz = float(m.group(1))
if sq and clock_mhz and "SQ_INSTS_VALU" in sq and "SQ_LDS_IDX_ACTIVE" in sq:
    CUS, SIMDS = 256, 1024
    sec = {
        "config": config, "kernel": summary.get("kernel", KERNEL), "clock_mhz": clock_mhz,
        # a wave's VALU instruction occupies its SIMD's issue for 4 cycles (64 lanes on 16); summed over all waves / 1024 SIMDs
        "valu_issue_ms": sq["SQ_INSTS_VALU"] * 4.0 / SIMDS / (clock_mhz * 1e3),
        # cycles a CU's LDS was serving indexed operations, summed over CUs / 256 CUs
        "lds_active_ms": sq["SQ_LDS_IDX_ACTIVE"] / CUS / (clock_mhz * 1e3),
        "lds_conflict_ratio": sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_LDS_IDX_ACTIVE"],
        "counters": {k: sq[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_LDS_IDX_ACTIVE",
                                        "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES") if k in sq},
        "kernel_ms_under_rocprofv3": summary.get("avg_ns", 0.0) * 1e-6,
        "source": "rocprofv3 --pmc SQ_* passes of scripts/collect_profiles.sh %s %s (per launch, summed over XCDs), shader clock from "
                  "rocm-smi while the same workload ran un-profiled; not measured by the bench run that quotes it" % (tag, config),
    }
    json.dump(sec, open(os.path.join(out, "secondary_%s.json" % config), "w"), indent=1)
    if "hbm_bytes_per_launch" in summary:
        json.dump({k: summary[k] for k in ("fetch_size_kb_per_launch", "write_size_kb_per_launch", "hbm_bytes_per_launch")} |
                  {"kernel": summary.get("kernel", KERNEL), "note": "scripts/collect_profiles.sh %s %s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                   "separate passes; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 16-byte-per-lane reads at 1/2, "
                   "MI355X_MICROARCH.md HBM section)" % (tag, config)},
                  open(os.path.join(out, "traffic_%s.json" % config), "w"), indent=1)
json.dump(summary, open(os.path.join(out, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary))
