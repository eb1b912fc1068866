"""Multi-GPU pre-flight on one GPU (-m gpu): the N > 1 path of bench.py / cli.py (RCCL process group, scatter + all-gather weight
broadcast, graph capture beside the RCCL watchdog, output all-gather) at world size 1 must reproduce the plain run bit for bit.
(The world-2 logic - sharding, ragged gathers, both broadcast forms - is covered on CPU/gloo by tests/test_dist_cpu.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_force_dist_world1_matches_plain_run():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    p = subprocess.run([sys.executable, os.path.join(HERE, "dist_gpu_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]          # (libdrm prints a notice line of its own)
    assert lines, (p.stdout[-1000:], p.stderr[-2000:])
    v = json.loads(lines[-1])
    assert v["moved"] > 1_000_000 and v["finite"], v
    assert v["same_as_plain"] and v["replay_same"] and v["plain_replay_same"], v


def test_bench_force_dist_through_the_driver_style_launch():
    """The driver's N > 1 command line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
    bench.py --gpus N`) at N = 1 with --force-dist: RCCL init, weight scatter + all-gather, graph capture beside the watchdog, the
    output all-gather inside the timed step and the MAX all-reduce of the timings run end to end; the JSON line carries the
    compute_ms / allgather_ms split that the N > 1 line carries.  (No scaling curve has been measured: there is one GPU per box here.)"""
    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29551", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--force-dist",
           "--no-cpu-baseline", "--no-other-configs", "--no-fp16", "--no-profile"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, (p.stdout[-1000:], p.stderr[-2000:])
    v = json.loads(lines[-1])
    assert v["n_gpus"] == 1 and v["steps"] == 2 and v["output_finite"] and v["scaling"] == "weak", v
    assert v["compute_ms"] > 0 and v["allgather_ms"] >= 0 and v["compute_ms"] + v["allgather_ms"] <= 1.05 * v["ms_per_step"] + 1.0, v
    assert abs(v["value"] - 8 * 1e3 / v["ms_per_step"]) < 1e-6 * v["value"] + 1e-9, v
