cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04d; O=$GRAFT_REPO_ROOT/gpurun_out/r04d
export HSA_ENABLE_COREDUMP=0 
ulimit -c 0
t() { tag=$1; shift
  env "$@" timeout 60 python -m pytest tests/test_eedi2_gpu.py -x -q -m gpu -k "test_every_scratch_buffer" > $O/pytest_$tag.log 2>&1; echo "$tag rc=$? $(tail -1 $O/pytest_$tag.log)"
}
t tile3 HBHIP_EEDI2_CALCDIR_TILE3=1
t s1 HBHIP_EEDI2_CALCDIR_SORT=1
t s0 HBHIP_EEDI2_CALCDIR_SORT=0
t s2 HBHIP_EEDI2_CALCDIR_SORT=2
