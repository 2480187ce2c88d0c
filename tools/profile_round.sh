set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01d_trace -o t -- python3 $R/bench.py --steps 100 --no-cpu > $R/gpurun_out/r01d_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01d_fetch -o f -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu > $R/gpurun_out/r01d_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01d_write -o w -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu > $R/gpurun_out/r01d_write.log 2>&1
cd $R
python3 profiles/summarize.py gpurun_out/r01d_trace gpurun_out/r01d_fetch gpurun_out/r01d_write "round 1, end of round (packed-fp32 bilinear blend, 2-stage SAMPLE_LDS tiles, per-tile table); commands: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --no-cpu ; rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 20 --warmup 2 --no-cpu" gpurun_out/r01d_pmc.json 110,60,60,60 22,12,12,12 > gpurun_out/r01_d_end_of_round.txt
cat gpurun_out/r01_d_end_of_round.txt
python3 bench.py > gpurun_out/r01d_bench_line.json 2> gpurun_out/r01d_bench.err
cat gpurun_out/r01d_bench_line.json
