#!/bin/bash
# CPU only (no GPU needed): builds oracle/tt_cpu.c with AddressSanitizer + UBSan and runs the CPU twin / coarse-entry tests on it.
# usage: bash tools/asan_twins.sh       (restores the normal build afterwards)
set -e
cd "$(dirname "$0")/.."
python -c "from oracle import cpu_twin; cpu_twin.build()"
cp oracle/_build/libtt_cpu.so /tmp/libtt_cpu.orig.so
gcc -O1 -g -std=c11 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -Wall -Wno-unused-parameter oracle/tt_cpu.c -o oracle/_build/libtt_cpu.so -lm
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_cpu_twin.py tests/test_coarse_entries.py -x -q -m "not gpu" || rc=$?
cp /tmp/libtt_cpu.orig.so oracle/_build/libtt_cpu.so
touch oracle/_build/libtt_cpu.so
exit ${rc:-0}
