#!/bin/bash
# One kernel file rebuilt with other -D flags, linked with the product's objects into build/<name>/libhbhip.so
# (for tools/dev_run.sh: DEVLIB=build/<name>/libhbhip.so).  usage: tools/variant.sh <name> <file.hip> [-Dflag ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
name=$1; src=$2; shift 2
mkdir -p build/$name
base=$(basename "$src" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -Ihandbrake_amd/csrc -Wall -Wno-unused-function "$@" -c "$src" -o build/$name/$base.o
objs=""
for o in handbrake_amd/csrc/*.o; do
    if [ "$(basename $o .o)" = "$base" ]; then objs="$objs build/$name/$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/$name/libhbhip.so $objs
echo "build/$name/libhbhip.so"
