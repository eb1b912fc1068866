#!/bin/bash
# round profile: kernel stats of bench.py, per-shape table, PMC passes of the dominant conv launch.  usage: tools/run_prof.sh <tag>
TAG=${1:-r3_b}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $TAG -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-fp16 > gpurun_out/${TAG}_prof_bench.json 2> gpurun_out/${TAG}_prof.err
f=$(ls gpurun_out/prof/*kernel_stats.csv gpurun_out/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
python tools/summarize_rocprof.py $f gpurun_out/${TAG}_rocprofv3_kernel_stats.csv 40
rm -rf gpurun_out/prof
timeout 600 python tools/prof_shapes.py 110 > gpurun_out/${TAG}_per_shape_hip_events.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  echo "== $c" >> gpurun_out/${TAG}_pmc_halo_conv.txt
  timeout 600 bash tools/pmc_dom.sh "$c" >> gpurun_out/${TAG}_pmc_halo_conv.txt 2>&1
done
tail -30 gpurun_out/${TAG}_pmc_halo_conv.txt
head -45 gpurun_out/${TAG}_rocprofv3_kernel_stats.csv
