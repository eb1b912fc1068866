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
