# One box, one call: the bench line first (chip not yet warmed by profiling), then the rocprofv3 passes of the same build.
#   gpurun -- 'bash tools/refresh_round.sh r02 "label"'      then copy gpurun_out/r02_* into profiles/
set -e
TAG=${1:-r02}
LABEL=${2:-"round 2"}
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
bash tools/profile_round.sh $TAG "$LABEL" > gpurun_out/${TAG}_round.log 2>&1
bash tools/profile_plans.sh $TAG "$LABEL" > gpurun_out/${TAG}_plans.log 2>&1
bash tools/profile_file_pipeline.sh $TAG > gpurun_out/${TAG}_file_pipeline.log 2>&1 || echo "file pipeline trace failed (see gpurun_out/${TAG}_file_pipeline.log)"
python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["kernel_us"], d["roofline"]["frac"], [d["extra"][k]["kernel_us"] for k in ("uniform_horizontal", "mixed_vertical", "mixed_horizontal")], d["d2d_copy_yardstick"]["us"])
PY
grep "^uniform\|^mixed" gpurun_out/${TAG}_summary.txt | head -4 | cut -c1-150
