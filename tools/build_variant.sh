#!/bin/bash
# Builds a variant of libtimetuning_hip.so with ONE translation unit recompiled under extra -D flags (A/B studies with
# tools/ab_attn.py / tools/ab_gemm.py).   usage: tools/build_variant.sh <name> <file.hip> [-DFLAG ...]
set -e
cd "$(dirname "$0")/../timetuning_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../../tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c "$src" -o "/tmp/variant_$name.o"
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "/tmp/variant_$name.o" -o "../../tools/bin/lib$name.so"
echo "built tools/bin/lib$name.so"
