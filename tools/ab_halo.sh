#!/bin/bash
# A/B libraries of the halo conv unit only: tools/ab_halo.sh "<tag>:<flags>" ...  -> unirestore_amd/ab/libur_<tag>.so (other objects from build/)
cd "$(dirname "$0")/.."
mkdir -p unirestore_amd/ab
OBJS=$(ls unirestore_amd/build/*.o | grep -v igemm_halo_bf16.o)
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $flags -DUR_TU_F16=0 -c unirestore_amd/csrc/igemm_halo.hip -o unirestore_amd/ab/halo_$tag.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unirestore_amd/ab/libur_$tag.so $OBJS unirestore_amd/ab/halo_$tag.o && echo built $tag ) &
done
wait
