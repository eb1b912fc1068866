#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
for t in ts ts_stag; do echo "== $t"; UR_LIB=$PWD/unirestore_amd/ab/libur_$t.so timeout 300 python tools/halo_ts.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/halo_ts.txt
