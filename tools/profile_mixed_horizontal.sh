# Counter-backed look at ONE workload: duration distribution from the kernel trace, then one --pmc
# counter per pass (never combined with a trace; the program directly after `--`).
#   gpurun -- 'bash tools/profile_mixed_horizontal.sh r03 mixed_horizontal'
set -e
TAG=${1:-r03}
WHICH=${2:-mixed_horizontal}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_mh_${WHICH}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/mh_workload.py --which $WHICH > $O/unprofiled.json 2> $O/unprofiled.err
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/tools/mh_workload.py --which $WHICH > $O/trace.json 2> $O/trace.err
for C in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -o c -- python3 $R/tools/mh_workload.py --which $WHICH --launches 10 --preroll 50 > $O/pmc_$C.json 2> $O/pmc_$C.err || echo "counter $C: pass failed" >> $O/failed.txt
done
cd $R
python3 tools/summarize_mh.py $O $WHICH > $R/gpurun_out/${TAG}_mh_${WHICH}.txt
cat $R/gpurun_out/${TAG}_mh_${WHICH}.txt
