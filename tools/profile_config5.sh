# BASELINE configs[4] (64 x 8000x6000 -> 8000x384000, one launch, 24.6 GB) under rocprofv3: kernel trace, then FETCH_SIZE and WRITE_SIZE in their
# own passes (never combined with a trace; the program directly after `--`).   gpurun -- 'bash tools/profile_config5.sh r04'
set -e
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_config5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--which config5 --sets 1 --preroll 20 --launches 10"
python3 $R/tools/mh_workload.py $ARGS > $O/unprofiled.json 2> $O/unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/tools/mh_workload.py $ARGS > $O/trace.json 2> $O/trace.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -o c -- python3 $R/tools/mh_workload.py --which config5 --sets 1 --preroll 5 --launches 5 > $O/pmc_$C.json 2> $O/pmc_$C.err || echo "counter $C: pass failed" >> $O/failed.txt
done
cd $R
python3 - <<PY > gpurun_out/${TAG}_config5.txt
import csv, glob, json, statistics
O = "$O"
print("# BASELINE configs[4] on one GPU: 64 x 8000x6000 -> 8000x384000, ONE launch (tools/profile_config5.sh; tools/mh_workload.py --which config5 --sets 1 --preroll 20 --launches 10)")
print("un-profiled:", open(O + "/unprofiled.json").read().strip())
rows = [r for f in glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "ist_stitch_kernel" in r["Kernel_Name"])
last = d  # all launches of the run (pre-roll + timed) are the same job
print("rocprofv3 --kernel-trace: %d launches of ist_stitch_kernel, us: min %.1f median %.1f mean %.1f max %.1f" % (len(d), d[0], statistics.median(d), statistics.mean(d), d[-1]))
B = json.loads(open(O + "/unprofiled.json").read().strip().splitlines()[-1])["algorithmic_bytes"]
print("algorithmic bytes per launch %d -> %.3f of 8 TB/s at the median" % (B, B / (statistics.median(d) * 1e-6) / 8e12))
for C, mul, note in (("FETCH_SIZE", 2.0, "x2: on gfx950 FETCH_SIZE reports half of a 16-B/lane stream (MI355X_MICROARCH.md, HBM section)"), ("WRITE_SIZE", 1.0, "exact for 16-B/lane streaming stores")):
    vals = []
    for f in glob.glob(O + "/pmc_" + C + "/**/*counter_collection.csv", recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if "ist_stitch_kernel" in r["Kernel_Name"] and r["Counter_Name"] == C:
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        vals += list(per.values())
    if vals:
        kb = statistics.mean(vals)
        print("--pmc %s: %d dispatches, mean %.1f KB -> %.2f MB per launch (%s) = %.4f of the %.2f MB this side moves" % (C, len(vals), kb, kb * 1024 * mul / 1e6, note, kb * 1024 * mul / (B / 2), B / 2 / 1e6))
PY
cat gpurun_out/${TAG}_config5.txt
