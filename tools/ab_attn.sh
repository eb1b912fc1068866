#!/bin/bash
# A/B libraries of the ping-pong attention unit: tools/ab_attn.sh "<tag>:<ATTN_ABL>:<extra hipcc flags>" ...
#   -> unirestore_amd/ab/libur_attn_<tag>.so (other objects from build/); the generated block is rebuilt per variant
cd "$(dirname "$0")/.."
mkdir -p unirestore_amd/ab
OBJS=$(ls unirestore_amd/build/*.o | grep -v attention_pp_bf16.o)
for spec in "$@"; do
  tag=${spec%%:*}; rest=${spec#*:}; abl=${rest%%:*}; flags=${rest#*:}
  ( ATTN_DBG=${ATTN_DBG:-0} ATTN_ABL=$abl python tools/gen_attn_asm.py $PWD/unirestore_amd/ab/attn_$tag.inc >/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $flags -DUR_TU_F16=0 \
      "-DUR_ATTN_PP_INC=\"$PWD/unirestore_amd/ab/attn_$tag.inc\"" -c unirestore_amd/csrc/attention_pp.hip -o unirestore_amd/ab/attn_$tag.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unirestore_amd/ab/libur_attn_$tag.so $OBJS unirestore_amd/ab/attn_$tag.o && echo built $tag ) &
done
wait
