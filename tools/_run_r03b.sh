cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b; O=$GRAFT_REPO_ROOT/gpurun_out/r03b; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
PROF="python $R/bench.py --workload decomb_eedi2 --steps 4 --warmup 2 --no-cpu-baseline --no-pcie --no-kernel-timer"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_SQ -o pmc -- $PROF > $O/pmc_SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SMEM --output-format csv -d $O/pmc_SQ2 -o pmc -- $PROF > $O/pmc_SQ2.log 2>&1
cd $R
python tools/summarize_pmc.py $O $O/pmc_summary.json > /dev/null 2>&1
find $O -name '*kernel_trace.csv' -size +3M -delete
find $O -name '*counter_collection.csv' -delete
find $O -name '*.db' -delete
