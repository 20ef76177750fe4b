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
if sq:
    with open(os.path.join(out, "%s_sq_counters.csv" % tag), "w") as g:
        g.write("counter,mean_per_launch\n")
        for k in sorted(sq):
            g.write("%s,%.6g\n" % (k, sq[k]))
json.dump(summary, open(os.path.join(out, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary))
