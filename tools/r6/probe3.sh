#!/bin/bash
# Round 6, third GPU call: where does the time go - wstream per-wave time line; halo conv workgroup time lines + ablations on the Cout <= 256 shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
{
echo "== wstream timeline"
UR_LIB=unirestore_amd/ab/libur_wstl.so python tools/r6/wstream_timeline.py 1280 1280 8
UR_LIB=unirestore_amd/ab/libur_wstl.so python tools/r6/wstream_timeline.py 2560 1280 8
echo "== halo workgroup time lines (UR_HALO_ABL=7)"
for s in "128 128 512" "256 256 256" "320 320 64" "512 512 128"; do echo "-- $s"; UR_LIB=unirestore_amd/ab/libur_tl.so python tools/halo_wg_timeline.py $s; done
echo "== halo ablations: default / a1 no DMA waits / a2 no MFMA body / a3 prologue+epilogue only / a4 no epilogue / a5 no DMA in the loop"
for s in "128 128 512" "256 256 256" "320 320 64"; do
  python tools/r6/time_conv.py $s
  for a in a1 a2 a3 a4 a5; do UR_LIB=unirestore_amd/ab/libur_$a.so python tools/r6/time_conv.py $s; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/r6_c_timelines.txt
