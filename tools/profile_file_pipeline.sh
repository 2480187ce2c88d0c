# Kernel + memory-copy timeline of the file pipeline (nine 12 MP JPEGs -> PNG), one rocprofv3 run with tracing only (no --pmc).
#   gpurun -- 'bash tools/profile_file_pipeline.sh r03'   then copy gpurun_out/r03_file_pipeline_kernels.txt into profiles/
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_file_pipeline
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o t -- python3 $R/tools/exp_pipeline.py > $O/run.json 2> $O/run.err
cd $R
python3 tools/summarize_file_pipeline.py $O > gpurun_out/${TAG}_file_pipeline_kernels.txt
cat gpurun_out/${TAG}_file_pipeline_kernels.txt
