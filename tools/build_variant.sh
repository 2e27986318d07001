#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [-DDEFS...]: the library with ONE translation unit rebuilt under extra definitions,
# as variants/NAME.so (experiments on the GPU box copy it over sigdigger_amd/libsigdigger_amd.so)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p variants
python -m sigdigger_amd.build > /dev/null
base=$(basename "$src" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-value \
  -ffp-contract=off "$@" -c sigdigger_amd/csrc/$src -o variants/$name.$base.o
objs=$(ls sigdigger_amd/csrc/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so $objs variants/$name.$base.o -Wl,-rpath,/opt/rocm/lib -lpthread -ldl
echo variants/$name.so
