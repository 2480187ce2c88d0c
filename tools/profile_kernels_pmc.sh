# One --pmc counter per pass over a program, summarised per kernel (mean per dispatch of the LAST call's dispatches).
#   gpurun -- 'bash tools/profile_kernels_pmc.sh r03_huff "ist_jpeg" tools/exp_huff.py 3'
# $1 tag, $2 kernel-name filter, rest: the python program and its arguments (run directly behind `--`).
set -e
TAG=$1; FILTER=$2; shift 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in ${COUNTERS:-GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS}; do
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -o c -- python3 $R/"$@" > $O/pmc_$C.out 2> $O/pmc_$C.err || echo "counter $C: pass failed" >> $O/failed.txt
done
cd $R
python3 tools/summarize_kernels_pmc.py $O "$FILTER" > $R/gpurun_out/${TAG}_pmc.txt
cat $R/gpurun_out/${TAG}_pmc.txt
