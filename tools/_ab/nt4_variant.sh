#!/bin/bash
# tools/_ab/nt4_variant.sh NAME "-DCUM_NT4_PA=.. ..."  -> tools/_ab/lib_nt4_NAME.so (gemm.hip rebuilt with the flags, other objects reused)
set -e
cd /root/repo/cleanumamba_amd/csrc
d=/tmp/nt4v_$1; mkdir -p $d; cp ../../tools/_ab/build/*.o $d/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value -DCUM_AB -DCUM_NT4_ONLY_BIAS $2 -c gemm.hip -o $d/gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/lib_nt4_$1.so $d/*.o -lhipfft
echo built $1
