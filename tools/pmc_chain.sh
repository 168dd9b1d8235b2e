#!/bin/bash
# rocprofv3 kernel trace + PMC passes of one bench.py chain variant (serial: one stream, no EEDI2 fork), summarised the way
# tools/gpu_round.sh does for the default workload.  usage: tools/pmc_chain.sh <tag> [bench args...]
set -u
export HSA_ENABLE_COREDUMP=0; ulimit -c 0
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PROF="env HBHIP_EEDI2_FORK=0 python $R/bench.py --workload chain --stage-streams 0 --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-kernel-timer $*"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $PROF > $OUT/kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- $PROF > $OUT/pmc_$C.log 2>&1
done
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_SQ -o pmc -- $PROF > $OUT/pmc_SQ.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_SQ2 -o pmc -- $PROF > $OUT/pmc_SQ2.log 2>&1
cd $R
python tools/summarize_pmc.py $OUT $OUT/pmc_summary.json > /dev/null 2>&1
python tools/kernel_bounds.py $OUT/pmc_summary.json $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_bounds.json > $OUT/kernel_bounds.txt 2>&1
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*counter_collection.csv' -delete
find $OUT -name '*.db' -delete
cat $OUT/kernel_bounds.txt | head -60
