cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -u -m pytest tests -v -m gpu --timeout=600 --durations=8 > gpurun_out/t_final.log 2>&1; echo "pytest rc=$?" > gpurun_out/t_final.rc
tail -15 gpurun_out/t_final.log
python bench.py > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err; echo "bench rc=$?"
head -c 600 gpurun_out/bench_r2c.json
