#!/bin/bash
# scratch GPU script
cd /root/repo/tools/probe
for f in p2_*; do timeout 60 ./$f 2>&1 | grep -v amdgpu.ids; done | tee /root/repo/gpurun_out/probe2.txt
