#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + PMC passes) of `bench.py` into a small text file for profiles/.
usage: summarize.py <trace_dir> <fetch_dir> <write_dir> <label>"""
import collections
import csv
import glob
import sys


def rows(d, pat):
    f = glob.glob(d + "/**/" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def main():
    trace, fetch, write, label = sys.argv[1:5]
    out = ["# rocprofv3 summary — %s" % label, ""]
    ks = rows(trace, "*_kernel_stats.csv")
    out.append("## --kernel-trace --stats (all dispatches of the run)")
    for r in ks:
        out.append("%-45s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (r["Name"][:45], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
    kt = [r for r in rows(trace, "*_kernel_trace.csv") if "ist_stitch" in r["Kernel_Name"]]
    # bench.py runs its configs back to back; group consecutive dispatches by grid size + order
    groups, cur = [], None
    for r in kt:
        key = r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?")
        if cur is None or cur[0] != key:
            cur = [key, []]
            groups.append(cur)
        cur[1].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out += ["", "## per bench configuration (consecutive dispatches with the same grid), durations in us"]
    for key, d in groups:
        d2 = sorted(d)
        out.append("grid=%s n=%d avg=%.2f median=%.2f min=%.2f max=%.2f" % (key, len(d), sum(d) / len(d) / 1e3, d2[len(d2) // 2] / 1e3, d2[0] / 1e3, d2[-1] / 1e3))
    for name, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        agg = collections.defaultdict(list)
        for r in rows(d, "*_counter_collection.csv"):
            if "ist_stitch" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            mean = sum(v) / len(v)
            note = " (x2 on gfx950 for 16-B/lane streams per MI355X_MICROARCH.md -> %.1f MB)" % (2 * mean * 1024 / 1e6) if k == "FETCH_SIZE" else " (-> %.1f MB)" % (mean * 1024 / 1e6)
            out.append("")
            out.append("## --pmc %s (headline config only): n=%d mean=%.1f KB min=%.1f max=%.1f%s" % (k, len(v), mean, min(v), max(v), note))
    print("\n".join(out))


if __name__ == "__main__":
    main()
