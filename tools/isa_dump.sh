#!/bin/bash
# Disassembly of the device code of one .hip unit (gfx950), one file per kernel: the tool behind
# "the refactor left the ISA of k_permute_lists untouched" checks.   tools/isa_dump.sh <unit.hip> <outdir>
set -e
cd "$(dirname "$0")/.."
SRC=${1:-scoary_amd/csrc/scoary_lists.hip}; OUT=${2:-/tmp/isa}
LLVM=${ROCM_PATH:-/opt/rocm}/lib/llvm/bin
mkdir -p "$OUT"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude --cuda-device-only -c -o "$OUT/unit.o" "$SRC" 2>/dev/null
$LLVM/clang-offload-bundler --unbundle --type=o --input="$OUT/unit.o" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$OUT/unit.gfx950.o"
$LLVM/llvm-objdump -d "$OUT/unit.gfx950.o" > "$OUT/unit.dis"
python3 - "$OUT" <<'PY'
import re, sys, hashlib
out = sys.argv[1]
name, body, res = None, [], {}
for line in open(out + "/unit.dis"):
    m = re.match(r"[0-9a-f]+ <(\S+)>:", line)
    if m:
        if name: res[name] = body
        name, body = m.group(1), []
        continue
    if name:
        # drop addresses and encodings: mnemonic + operands only
        t = line.split("//")[0].strip()
        if t: body.append(re.sub(r"\s+", " ", t))
if name: res[name] = body
for k, b in sorted(res.items()):
    print("%s  %6d instr  %s" % (hashlib.sha256("\n".join(b).encode()).hexdigest()[:16], len(b), k))
PY
