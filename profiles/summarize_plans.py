#!/usr/bin/env python3
"""Joins tools/pmc_workloads.py's list with the rocprofv3 CSVs of its three passes (kernel trace, FETCH_SIZE, WRITE_SIZE).
usage: summarize_plans.py <workloads.jsonl> <trace_dir> <fetch_dir> <write_dir> <label>"""
import csv
import glob
import json
import sys


def rows(d, pat):
    f = glob.glob(d + "/**/" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def main():
    wl = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
    trace, fetch, write, label = sys.argv[2:6]
    kt = [r for r in sorted(rows(trace, "*_kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"])) if "ist_stitch" in r["Kernel_Name"]]
    fc = [r for r in sorted(rows(fetch, "*_counter_collection.csv"), key=lambda r: int(r["Start_Timestamp"])) if "ist_stitch" in r["Kernel_Name"]]
    wc = [r for r in sorted(rows(write, "*_counter_collection.csv"), key=lambda r: int(r["Start_Timestamp"])) if "ist_stitch" in r["Kernel_Name"]]
    print("# rocprofv3 summary of the supplementary plans - %s" % label)
    print("# FETCH_SIZE x2 (gfx950 reports half of a wide streaming read, MI355X_MICROARCH.md HBM section); this correction is calibrated for 16-B/lane streams,")
    print("# so for the 4/8-byte gathers of the direct SAMPLE path (nearest plans) the fetched figure is an upper estimate. WRITE_SIZE as reported.")
    print("%-78s %9s %10s %10s %10s %8s %8s" % ("workload", "avg us", "algo MB", "fetch MB", "write MB", "traf/algo", "frac@8TB/s"))
    i = 0
    for w in wl:
        n = w["launches"]
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt[i:i + n]]
        f = [float(r["Counter_Value"]) for r in fc[i:i + n]]
        o = [float(r["Counter_Value"]) for r in wc[i:i + n]]
        i += n
        if not d:
            continue
        us = sum(d[2:]) / len(d[2:]) / 1e3          # the first two launches warm the caches / clocks
        fb = 2.0 * sum(f) / len(f) * 1024 if f else float("nan")
        ob = sum(o) / len(o) * 1024 if o else float("nan")
        algo = w["algorithmic_bytes"]
        print("%-78s %9.1f %10.1f %10.1f %10.1f %8.2f %8.3f" % (w["workload"][:78], us, algo / 1e6, fb / 1e6, ob / 1e6, (fb + ob) / algo, algo / (us * 1e-6) / 8e12))


if __name__ == "__main__":
    main()
