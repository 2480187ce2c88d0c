# One box, one call: the bench line first (chip not yet warmed by profiling), then the rocprofv3 passes of the same build.
#   gpurun -- 'bash tools/refresh_round.sh r04 "label"'      then copy gpurun_out/r04_* into profiles/
set -e
TAG=${1:-r04}
LABEL=${2:-"round 4"}
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
# The mixed horizontal strip is the one configuration whose time depends on the box (113-132 us).  It is timed
# FIRST, un-profiled; when this box is one of the slow ones (>= 126 us) its counters are taken in THIS lease, beside the fast-box
# file of round 3 ("measure first, profile only if slow, in the same lease").
MH=$(python3 tools/mh_workload.py --which mixed_horizontal | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['event_us_per_launch'])")
MV=$(python3 tools/mh_workload.py --which mixed_vertical | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['event_us_per_launch'])")
echo "mixed_horizontal ${MH} us, mixed_vertical ${MV} us per launch (un-profiled, 400 launches of pre-roll, 50 timed)" | tee gpurun_out/${TAG}_mixed_first.txt
if python3 -c "import sys; sys.exit(0 if float('${MH}') >= 126.0 else 1)"; then
  echo "slow box for the horizontal strip: collecting its counters in this lease" | tee -a gpurun_out/${TAG}_mixed_first.txt
  bash tools/profile_mixed_horizontal.sh ${TAG}_slowbox mixed_horizontal > gpurun_out/${TAG}_mh_slowbox.log 2>&1 || echo "slow-box counter passes failed (see gpurun_out/${TAG}_mh_slowbox.log)"
  bash tools/profile_mixed_horizontal.sh ${TAG}_slowbox mixed_vertical > gpurun_out/${TAG}_mv_slowbox.log 2>&1 || true
fi
bash tools/profile_round.sh $TAG "$LABEL" > gpurun_out/${TAG}_round.log 2>&1
bash tools/profile_plans.sh $TAG "$LABEL" > gpurun_out/${TAG}_plans.log 2>&1
bash tools/profile_file_pipeline.sh $TAG > gpurun_out/${TAG}_file_pipeline.log 2>&1 || echo "file pipeline trace failed (see gpurun_out/${TAG}_file_pipeline.log)"
# the JPEG decode kernels (Huffman passes, chroma IDCT, fused reconstruction): durations per launch, then FETCH_SIZE / WRITE_SIZE per kernel
bash tools/profile_decode.sh $TAG > gpurun_out/${TAG}_decode.log 2>&1 || echo "decode trace failed (see gpurun_out/${TAG}_decode.log)"
COUNTERS="FETCH_SIZE WRITE_SIZE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" bash tools/profile_kernels_pmc.sh ${TAG}_reconstruct "ist_jpeg" tools/exp_huff.py 4 > gpurun_out/${TAG}_reconstruct_pmc.log 2>&1 || echo "decode counter passes failed"
python3 tools/exp_config5.py > gpurun_out/${TAG}_config5_widths.txt 2>&1 || true      # configs[4]: working set and row pitch (un-profiled)
bash tools/profile_config5.sh $TAG > gpurun_out/${TAG}_config5.log 2>&1 || echo "configs[4] passes failed (see gpurun_out/${TAG}_config5.log)"   # -> gpurun_out/${TAG}_config5.txt
python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["kernel_us"], d["roofline"]["frac"], [d["extra"][k]["kernel_us"] for k in ("uniform_horizontal", "mixed_vertical", "mixed_horizontal")], d["d2d_copy_yardstick"]["us"])
print("config5", d["extra"]["config5_single_gpu"].get("kernel_us"), d["extra"]["config5_single_gpu"].get("frac"))
fp = d["extra"]["file_pipeline"]; print("file pipeline", fp["ms_end_to_end"], fp["stages_ms"])
PY
grep "^uniform\|^mixed" gpurun_out/${TAG}_summary.txt | head -4 | cut -c1-150
