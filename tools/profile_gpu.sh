#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel-trace stats +
# PMC passes (each in its own run, --kernel-trace only, as gpurun requires).
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench.py "$@" --no-cpu-baseline > $OUT/kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_SQ -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_SQ.log 2>&1
find $OUT -name '*.csv' | head -30
# keep only small summaries
find $OUT -name '*kernel_trace.csv' -size +2M -delete
ls -la $OUT $OUT/kt 2>/dev/null | head -40
