#!/bin/bash
# AddressSanitizer + UBSan pass over the host-only native code (scoary_amd/csrc/scoary_io.cpp: gene-table reader,
# results writer, VCF record loop, UPGMA merge loop, host list builder) -- CPU build only (GPU sanitizers are not
# available on the pool).  Builds an instrumented library next to nothing in the tree (/tmp), points the package at it
# (SCOARY_IO_LIB) and runs the host tests + the byte-level mutation fuzz under it.
#   tools/sanitize_io.sh [fuzz cases] [seed]
set -eu
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/scoary_san
mkdir -p "$OUT"
g++ -O1 -g -std=c++17 -fPIC -shared -Wall -Wextra -fopenmp -ffp-contract=off -fsanitize=address,undefined \
    -fno-omit-frame-pointer -Iinclude scoary_amd/csrc/scoary_io.cpp -o "$OUT/libscoary_io_asan.so"
export SCOARY_IO_LIB="$OUT/libscoary_io_asan.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_host_logic.py tests/test_readers_fuzz.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -2
python tools/fuzz_reader_mutations.py "${1:-3000}" "${2:-1}" 2>&1 | grep -v "^\[" | tail -4
