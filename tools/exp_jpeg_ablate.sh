# Ablations of the fused JPEG reconstruction kernel (IST_TUNING=1 IST_JPEG_EXP=bits; 1: no global loads, 2: no IDCT arithmetic,
# 4: no colour stage / stores): kernel trace per variant.   gpurun -- 'bash tools/exp_jpeg_ablate.sh'
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for E in 0 1 2 4 3 6 5; do
  O=$R/gpurun_out/r04_jpeg_exp$E
  rm -rf $O; mkdir -p $O
  IST_TUNING=1 IST_JPEG_EXP=$E rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/tools/exp_huff.py 4 > $O/run.out 2> $O/run.err
  echo "exp $E: fused kernel us per launch (last call): $(python3 $R/tools/list_kernel_durations.py $O/trace ist_jpeg_fused_kernel 1000 | tail -1)"
done
