#!/bin/bash
# link_variant.sh <name>: librcfm.so from build_ab/<name>/*.o where present, the in-tree objects otherwise
set -e
cd /root/repo
L=radio-core_amd/radiocore/_lib
objs=""
for o in kernels fft_plan fft_engine fused_passes fused_passes_w8 fused_decim fused_decim_w8 lds_chain api; do
  if [ -f build_ab/$1/$o.o ]; then objs="$objs build_ab/$1/$o.o"; else objs="$objs $L/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $objs -shared -L/opt/rocm/lib -lrocfft -ldl -Wl,-rpath,/opt/rocm/lib -o build_ab/$1/librcfm.so
echo linked build_ab/$1/librcfm.so
