#!/bin/bash
# Runs a command with the development kernels library (build/dev/libhbhip.so, `make devlib`) in the product library's
# place and puts the product library back afterwards - also when the command is interrupted or killed by a timeout (the
# EXIT trap), so that later tests never run the HBHIP_DEV build unknowingly.  The backup has a name of its own per run.
# Usage: [DEVLIB=build/devX/libhbhip.so] tools/dev_run.sh <command ...>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R" || exit 2
bak=$(mktemp /tmp/libhbhip.prod.XXXXXX.so) || exit 2
cp handbrake_amd/libhbhip.so "$bak" || exit 2
trap 'cp "$bak" handbrake_amd/libhbhip.so; rm -f "$bak"' EXIT
trap 'exit 130' INT TERM
cp "${DEVLIB:-build/dev/libhbhip.so}" handbrake_amd/libhbhip.so || exit 2
"$@"
