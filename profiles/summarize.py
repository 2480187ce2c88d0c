#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + separate PMC passes) of `bench.py` into a small text file for
profiles/.  usage: summarize.py <trace_dir> <fetch_dir> <write_dir> <label> <out.json> <trace counts total:timed e.g. 510:100,460:50,460:50,460:50> <pmc counts e.g. 422:20,412:10,412:10,412:10>
bench.py runs its configurations back to back (uniform vertical, uniform horizontal, mixed vertical, mixed horizontal);
consecutive dispatches of the same kernel + grid are one configuration."""
import csv
import glob
import json
import sys

NAMES = ["uniform_vertical (BASELINE configs[1])", "uniform_horizontal (BASELINE configs[2])", "mixed_vertical (supplementary)", "mixed_horizontal (supplementary)"]


def rows(d, pat):
    f = glob.glob(d + "/**/" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def groups(rs, value, counts):
    """split the ist_stitch dispatches, in time order, into bench configurations of known dispatch counts
    (bench.py: pre-roll + warmup + steps per configuration); a count "total:timed" keeps only the last `timed` dispatches of
    the group (the timed steps, after the untimed pre-roll and warm-up)"""
    mine = [r for r in rs if "ist_stitch" in r["Kernel_Name"]]
    out, i = [], 0
    for c in counts:
        total, timed = (c, c) if isinstance(c, int) else c
        part = mine[i:i + total][-timed:]
        i += total
        if part:
            out.append([(part[0]["Kernel_Name"], part[0].get("Grid_Size") or part[0].get("Grid_Size_X")), [value(r) for r in part]])
    return out


def parse_counts(text):
    out = []
    for tok in text.split(","):
        a, _, b = tok.partition(":")
        out.append((int(a), int(b or a)))
    return out


def main():
    trace, fetch, write, label = sys.argv[1:5]
    out = ["# rocprofv3 summary — %s" % label, ""]
    out.append("## --kernel-trace --stats (all dispatches of the run, 4 bench configurations)")
    for r in rows(trace, "*_kernel_stats.csv"):
        out.append("%-60s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
    kt = sorted(rows(trace, "*_kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"]))
    out += ["", "## per bench configuration (kernel trace), durations in us"]
    machine = {}
    for i, (key, d) in enumerate(groups(kt, lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), parse_counts(sys.argv[6]))):
        d2 = sorted(d)
        name = NAMES[i] if i < len(NAMES) else "config %d" % i
        out.append("%-42s kernel=%s grid=%s n=%d avg=%.2f median=%.2f min=%.2f max=%.2f" % (name, key[0].split("(")[0][-40:], key[1], len(d), sum(d) / len(d) / 1e3, d2[len(d2) // 2] / 1e3, d2[0] / 1e3, d2[-1] / 1e3))
        machine[name] = {"avg_us": sum(d) / len(d) / 1e3, "median_us": d2[len(d2) // 2] / 1e3, "n": len(d)}
    for cname, d, scale, note in (("FETCH_SIZE", fetch, 2.0, "x2: on gfx950 FETCH_SIZE reports 1/2 of a 16-B/lane stream, MI355X_MICROARCH.md HBM section"), ("WRITE_SIZE", write, 1.0, "exact for 16-B/lane streaming stores")):
        cc = sorted(rows(d, "*_counter_collection.csv"), key=lambda r: int(r["Start_Timestamp"]))
        out += ["", "## --pmc %s (own pass; KB per dispatch; %s)" % (cname, note)]
        for i, (key, v) in enumerate(groups(cc, lambda r: float(r["Counter_Value"]), parse_counts(sys.argv[7]))):
            name = NAMES[i] if i < len(NAMES) else "config %d" % i
            mean = sum(v) / len(v)
            out.append("%-42s n=%d mean=%.1f KB -> %.2f MB per launch" % (name, len(v), mean, scale * mean * 1024 / 1e6))
            machine.setdefault(name, {})[cname + "_bytes"] = scale * mean * 1024
    print("\n".join(out))
    if len(sys.argv) > 5:
        # which kernel + tiling these numbers belong to: bench.py quotes roofline.traffic only when this matches its own hash
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        machine["kernel_source_sha"] = bench.kernel_source_sha()
        json.dump(machine, open(sys.argv[5], "w"), indent=1)


if __name__ == "__main__":
    main()
