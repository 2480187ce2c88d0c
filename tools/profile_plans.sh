# rocprofv3 passes of the supplementary plans (phone-capped plans, scaled EXIF quarter turns): kernel trace, then FETCH_SIZE and
# WRITE_SIZE each in its own pass.  gpurun -- 'bash tools/profile_plans.sh r02 "label"'
set -e
TAG=${1:-r02}
LABEL=${2:-"round 2"}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_plans_trace -o t -- python3 $R/tools/pmc_workloads.py > $R/gpurun_out/${TAG}_plans.jsonl 2> $R/gpurun_out/${TAG}_plans_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_plans_fetch -o f -- python3 $R/tools/pmc_workloads.py > /dev/null 2> $R/gpurun_out/${TAG}_plans_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_plans_write -o w -- python3 $R/tools/pmc_workloads.py > /dev/null 2> $R/gpurun_out/${TAG}_plans_write.err
cd $R
python3 profiles/summarize_plans.py gpurun_out/${TAG}_plans.jsonl gpurun_out/${TAG}_plans_trace gpurun_out/${TAG}_plans_fetch gpurun_out/${TAG}_plans_write "$LABEL" > gpurun_out/${TAG}_plans_summary.txt
cat gpurun_out/${TAG}_plans_summary.txt
