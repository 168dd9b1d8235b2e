#!/bin/bash
# Device code of the product kernels library must not be able to abort the process: no s_trap in any gfx950 code object
# of handbrake_amd/libhbhip.so (tests/test_abi.py runs this when the ROCm binutils are present).
# usage: tools/no_trap_check.sh [library]   -> prints the number of s_trap instructions, exit 1 when there are any
set -e
LIB=${1:-handbrake_amd/libhbhip.so}
BIN=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
$BIN/clang-offload-bundler --list --type=o --input="$LIB" >/dev/null 2>&1 || true
# the fat binary section holds the bundle: unbundle the gfx950 code objects and disassemble them
$BIN/llvm-objcopy --dump-section .hip_fatbin="$T/fat.bin" "$LIB" 2>/dev/null
n=0
python3 - "$T/fat.bin" "$T" <<'PY'
import sys, struct
d = open(sys.argv[1], 'rb').read()
out = sys.argv[2]
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
k = 0
pos = d.find(MAGIC)
while pos >= 0:
    nb = struct.unpack_from('<Q', d, pos + 24)[0]
    p = pos + 32
    for _ in range(nb):
        off, size, tlen = struct.unpack_from('<QQQ', d, p)
        triple = d[p + 24:p + 24 + tlen].decode()
        p += 24 + tlen
        if 'gfx950' in triple and size:
            open(f'{out}/co{k}.o', 'wb').write(d[pos + off:pos + off + size])
            k += 1
    pos = d.find(MAGIC, pos + 1)
print(k, 'gfx950 code objects', file=sys.stderr)
PY
for f in "$T"/co*.o; do
    c=$($BIN/llvm-objdump -d --mcpu=gfx950 "$f" | grep -c 's_trap' || true)
    n=$((n + c))
done
echo "$n"
[ "$n" -eq 0 ]
