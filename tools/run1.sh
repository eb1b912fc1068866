#!/bin/bash
# scratch GPU script
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_chain_gpu.py -x -q 2>&1 | tail -8
timeout 900 python -m pytest tests/test_modules_gpu.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench4.json 2> gpurun_out/bench4.err; tail -c 3000 gpurun_out/bench4.json
