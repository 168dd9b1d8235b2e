cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02g; O=$GRAFT_REPO_ROOT/gpurun_out/r02g
timeout 900 python -m pytest tests/test_eedi2_gpu.py tests/test_configs_gpu.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie > $O/bench_decomb.json 2> $O/bench_decomb.err
python - <<PY
import json
b=json.load(open("$O/bench_decomb.json"))
print(b["value"], [(k["kernel"].replace("eedi2_",""), k["avg_us"]) for k in b["kernels"]])
PY
export TMPDIR=/tmp; cd /tmp
PROF="python $GRAFT_REPO_ROOT/bench.py --workload decomb_eedi2 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-kernel-timer"
HBHIP_EEDI2_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_SQ -o pmc -- $PROF > $O/pmc_SQ.log 2>&1
HBHIP_EEDI2_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/pmc_SQ2 -o pmc -- $PROF > $O/pmc_SQ2.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/summarize_pmc.py $O $O/pmc_summary.json > /dev/null 2>&1
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -size +3M -delete; find $O -name '*.db' -delete
