cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -8
python tools/bench_chain.py 2>&1 | tail -2
timeout 900 python bench.py --steps 3 --warmup 1 2>gpurun_out/bench_chain.err | tail -1 > gpurun_out/bench_chain.json; cat gpurun_out/bench_chain.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:(v['launches'], round(v['ms'],1)) for k,v in d['families'].items()})"
UR_CHAIN=0 timeout 900 python bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('UR_CHAIN=0', d['value'], d['ms_per_step']); print({k:(v['launches'], round(v['ms'],1)) for k,v in d['families'].items()})"
