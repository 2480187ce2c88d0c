# Kernel trace of the GPU JPEG decode (nine 12 MP photos -> bitmaps in HBM): per-kernel durations of the Huffman and
# reconstruction kernels.   gpurun -- 'bash tools/profile_decode.sh r04'
set -e
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_decode
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/tools/exp_huff.py 6 > $O/run.out 2> $O/run.err
cd $R
F=$(ls $O/trace/*/*_kernel_stats.csv $O/trace/*_kernel_stats.csv 2>/dev/null | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 tools/exp_huff.py 6   (nine 12 MP photo-like JPEGs -> bitmaps in HBM, 6 calls)"
  cat "$F"
  for k in ist_jpeg_idct_kernel ist_jpeg_fused_kernel ist_jpeg_sync_kernel ist_jpeg_write_kernel; do
    echo "# durations (us) of $k in launch order, one line per call:"
    python3 tools/list_kernel_durations.py $O/trace $k 1000
  done
} > gpurun_out/${TAG}_decode_kernels.txt
cat gpurun_out/${TAG}_decode_kernels.txt
