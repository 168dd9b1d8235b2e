cd $GRAFT_REPO_ROOT; timeout 600 python tools/host_path_probe.py 2>&1 | grep variant
