# rocprofv3 passes of the bench's kernel configurations (run on the GPU box: gpurun -- 'bash tools/profile_round.sh r02a "label"')
# kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their OWN passes (never combined with tracing domains).
set -e
TAG=${1:-r02}
LABEL=${2:-"round 2"}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -o t -- python3 $R/bench.py --steps 100 --kernels-only > $R/gpurun_out/${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -o f -- python3 $R/bench.py --steps 20 --warmup 2 --kernels-only > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -o w -- python3 $R/bench.py --steps 20 --warmup 2 --kernels-only > $R/gpurun_out/${TAG}_write.log 2>&1
cd $R
python3 profiles/summarize.py gpurun_out/${TAG}_trace gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write "$LABEL; commands: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --kernels-only ; rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 20 --warmup 2 --kernels-only" gpurun_out/${TAG}_pmc.json 510:100,460:50,460:50,460:50 422:20,412:10,412:10,412:10 > gpurun_out/${TAG}_summary.txt
cat gpurun_out/${TAG}_summary.txt
