"""Worker for tests/test_dist_cpu.py: one gloo rank; prints a JSON verdict."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unirestore_amd import dist as ud
    torch.manual_seed(100 + rank)                       # different init per rank; broadcast must make them equal
    model = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.BatchNorm1d(19), torch.nn.Linear(19, 5))
    moved = ud.broadcast_weights(model, src=0, bucket_bytes=512)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    full = torch.arange(7 * 3 * 2 * 2, dtype=torch.float32).view(7, 3, 2, 2)
    lo, hi = ud.shard_range(7, rank, world)
    sizes = [ud.shard_range(7, r, world)[1] - ud.shard_range(7, r, world)[0] for r in range(world)]
    ragged = ud.all_gather_images(full[lo:hi] * 1.0, sizes)       # ragged shards keep batch order
    even = ud.all_gather_images(full[:6][rank * 3:(rank + 1) * 3] * 1.0)
    print(json.dumps(dict(rank=rank, same=same, moved=moved, ragged_ok=torch.equal(ragged, full), even_ok=torch.equal(even, full[:6]))))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
