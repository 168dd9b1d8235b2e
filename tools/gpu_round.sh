#!/bin/bash
# One GPU-box session (via gpurun): GPU parity tests, micro-benchmarks, bench lines, rocprofv3 kernel
# trace + PMC passes of the default bench workload.  Everything lands under gpurun_out/<tag>/.
# Usage: tools/gpu_round.sh <tag> [tests|notests] [extra pytest args...]
set -u
export HSA_ENABLE_COREDUMP=0; ulimit -c 0
TAG=${1:-r02z}; MODE=${2:-tests}; shift 2 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
if [ "$MODE" = "tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q -n 4 "$@" > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
# (valu_rate: run once per round, profiles/r02_valu_rate.json)
# the driver's command line first (defaults), then the other workloads
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
head -c 1200 $OUT/bench_default.json; echo
for C in corners random; do
  timeout 300 python bench.py --content $C --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_chain_$C.json 2> $OUT/bench_chain_$C.err
  head -c 400 $OUT/bench_chain_$C.json; echo
done
for WL in chain2160 decomb_eedi2 nlmeans; do
  timeout 240 python bench.py --workload $WL --steps 20 --warmup 3 --no-pcie > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  head -c 600 $OUT/bench_$WL.json; echo
done
timeout 200 python bench.py --workload decomb_eedi2 --depth 10 --no-cpu-baseline --no-pcie > $OUT/bench_decomb_eedi2_10bit.json 2> $OUT/bench_decomb_eedi2_10bit.err
timeout 200 python bench.py --workload chain --depth 10 --no-cpu-baseline --no-pcie > $OUT/bench_chain_10bit.json 2> $OUT/bench_chain_10bit.err
timeout 200 python bench.py --workload nlmeans --depth 10 --no-cpu-baseline --no-pcie > $OUT/bench_nlmeans_10bit.json 2> $OUT/bench_nlmeans_10bit.err
timeout 200 python bench.py --workload chain --stage-streams 1 --no-cpu-baseline --no-pcie --no-kernel-timer > $OUT/bench_chain_stage_streams.json 2>> $OUT/bench_default.err
timeout 200 python bench.py --workload chain --stage-streams 0 --no-cpu-baseline --no-pcie --no-kernel-timer > $OUT/bench_chain_one_stream.json 2>> $OUT/bench_default.err
timeout 200 python bench.py --workload chain --streams 2 --stage-streams 0 --no-cpu-baseline --no-pcie --no-kernel-timer > $OUT/bench_chain_2streams.json 2>> $OUT/bench_default.err
timeout 200 python bench.py --workload chain --comb-detect --no-cpu-baseline --no-pcie --no-kernel-timer > $OUT/bench_chain_combdetect.json 2>> $OUT/bench_default.err
timeout 240 python tools/kernel_rooflines.py > $OUT/kernel_rooflines.json 2> $OUT/kernel_rooflines.err
# the same table on 2160p frames: the 16 frames of a batched launch (199 MB in, 199 MB out) are past the 256 MB Infinity Cache
KR_W=3840 KR_H=2160 timeout 400 python tools/kernel_rooflines.py > $OUT/kernel_rooflines_2160p.json 2> $OUT/kernel_rooflines_2160p.err
cd /tmp
PROF="env HBHIP_EEDI2_FORK=0 python $R/bench.py --workload chain --stage-streams 0 --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-kernel-timer"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $PROF > $OUT/kt.log 2>&1
python $R/tools/trace_gaps.py $(find $OUT/kt -name '*kernel_trace.csv' | head -1) $OUT/trace_gaps.json > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- $PROF > $OUT/pmc_$C.log 2>&1
done
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_SQ -o pmc -- $PROF > $OUT/pmc_SQ.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_SQ2 -o pmc -- $PROF > $OUT/pmc_SQ2.log 2>&1
cd $R
python tools/summarize_pmc.py $OUT $OUT/pmc_summary.json > /dev/null 2>&1
python tools/kernel_bounds.py $OUT/pmc_summary.json $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_bounds.json > $OUT/kernel_bounds.txt 2>&1
# the forked production path's trace too (what the chain really runs: EEDI2's side streams, decomb beside the other stages)
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_fork -o kt -- python $R/bench.py --workload chain --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-kernel-timer > $OUT/kt_fork.log 2>&1
cd $R
python tools/trace_overlap.py $(find $OUT/kt_fork -name '*kernel_trace.csv' | head -1) $OUT/fork_overlap.json 250 > /dev/null 2>&1
# the 10-bit chain under the same counters (tools/pmc_chain.sh)
tools/pmc_chain.sh $TAG/chain10 --depth 10 > $OUT/chain10_bounds.txt 2>&1
# keep only small files
find $OUT -name '*kernel_trace.csv' -size +3M -delete
find $OUT -name '*counter_collection.csv' -delete
find $OUT -name '*.db' -delete
du -sh $OUT
