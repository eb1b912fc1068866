"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: weight broadcast, batch sharding, output all-gather."""
import json
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_broadcast_shard_gather_world2():
    port, world = _free_port(), 2
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        v = json.loads(out.strip().splitlines()[-1])
        assert v["same"] and v["moved"] > 0 and v["ragged_ok"] and v["even_ok"], v
        assert v["tree_same"] and v["tree_bytes"] > 1_000_000 and v["forms_agree"] and v["data_ok"], v


def test_shard_range_covers_everything():
    from unirestore_amd.dist import shard_range
    for n in (0, 1, 7, 8, 64):
        for w in (1, 2, 4, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
