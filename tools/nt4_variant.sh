#!/bin/bash
# tools/nt4_variant.sh NAME "-DCUM_NT4_PA=.. ..."  -> tools/_ab/lib_nt4_NAME.so: gemm.hip rebuilt with -DCUM_AB -DCUM_NT4_ONLY_BIAS and
# the flags given, the other objects taken from an AB build in tools/_ab/build/ (make AB=1 objects copied there); CPU container.
set -e
cd /root/repo/cleanumamba_amd/csrc
d=/tmp/nt4v_$1; mkdir -p $d; cp ../../tools/_ab/build/*.o $d/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value -DCUM_AB -DCUM_NT4_ONLY_BIAS $2 -c gemm.hip -o $d/gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/lib_nt4_$1.so $d/*.o -lhipfft
echo built $1
