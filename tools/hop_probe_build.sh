#!/bin/bash
# library with the hop kernel's phase stamps compiled in: tools/_ab/lib_hop_probe.so  (CPU container; hipcc cross-compiles)
set -e
cd "$(dirname "$0")/../cleanumamba_amd/csrc"
mkdir -p ../../tools/_ab/obj
for f in *.hip; do
  o=../../tools/_ab/obj/${f%.hip}.o
  flags=""
  case $f in scan_bwd*.hip) flags="-fno-slp-vectorize";; esac
  if [ $f = hop.hip ]; then flags="-DCUM_HOP_PROBE ${HOP_PROBE_PC:+-DCUM_HOP_PROBE_PC=$HOP_PROBE_PC}"; o=../../tools/_ab/obj/hop_probe.o; fi
  if [ ! -f $o ] || [ $f -nt $o ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value $flags -c $f -o $o & fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/lib_hop_probe.so ../../tools/_ab/obj/*.o -lhipfft
