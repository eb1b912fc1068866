#!/bin/bash
# A/B library with every g1 tile shape x ring depth behind UR_AB_ID (see igemm_g1.hip): tools/ab_variants.sh -> unirestore_amd/build_ab/libur_ab.so
set -e
cd "$(dirname "$0")/.."
python -c "from unirestore_amd import build; build.build()"
mkdir -p unirestore_amd/build_ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -DUR_AB_VARIANTS"
/opt/rocm/bin/hipcc $F -DUR_TU_F16=0 -c unirestore_amd/csrc/igemm_g1.hip -o unirestore_amd/build_ab/igemm_g1_bf16.o &
/opt/rocm/bin/hipcc $F -c unirestore_amd/csrc/igemm.hip -o unirestore_amd/build_ab/igemm.o &
wait
OBJS=$(ls unirestore_amd/build/*.o | grep -v -e "/igemm_g1_bf16.o" -e "/igemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unirestore_amd/build_ab/libur_ab.so $OBJS unirestore_amd/build_ab/igemm_g1_bf16.o unirestore_amd/build_ab/igemm.o
echo built unirestore_amd/build_ab/libur_ab.so
