#!/bin/bash
# Build tools/_ab/lib_<name>.so = the current library with csrc/scan_bwd.hip recompiled under extra flags (same-box A/B of
# backward-scan experiments).   tools/build_scan_bwd_variant.sh <name> [-DFLAG ...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/cleanumamba_amd/csrc
mkdir -p $root/tools/_ab /tmp/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value -fno-slp-vectorize "$@" -c scan_bwd.hip -o /tmp/asm/scan_bwd_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/_ab/lib_$name.so $(ls build/*.o | grep -v "build/scan_bwd.o") /tmp/asm/scan_bwd_$name.o -lhipfft
echo built tools/_ab/lib_$name.so
