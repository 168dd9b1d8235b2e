#!/bin/bash
# Runs a command with the development kernels library (build/dev/libhbhip.so, `make devlib`) in the product library's
# place and puts the product library back afterwards.  Usage: [DEVLIB=build/devX/libhbhip.so] tools/dev_run.sh <command ...>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp handbrake_amd/libhbhip.so /tmp/libhbhip.prod.so && cp ${DEVLIB:-build/dev/libhbhip.so} handbrake_amd/libhbhip.so
"$@"; rc=$?
cp /tmp/libhbhip.prod.so handbrake_amd/libhbhip.so
exit $rc
