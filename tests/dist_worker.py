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
    # the real parameter + buffer tree of the path (tiny configuration): fp32 masters, the int64 timestep buffer, null_embeds
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tiny_cfg import TINY, model_kwargs, randomise_
    import unirestore_amd.modules as M
    tree = randomise_(M.DiffUIE(**model_kwargs(2), **TINY).eval(), 7 + rank)          # different values on every rank
    if rank != 0:
        tree.train_timesteps.zero_()
    import copy
    tree2 = copy.deepcopy(tree)                                                       # same pre-broadcast values on this rank
    n_bytes = ud.broadcast_weights_sharded(tree, src=0, bucket_bytes=1 << 20)         # scatter + all-gather form
    digest = torch.cat([t.detach().double().reshape(-1) for t in tree.state_dict().values()])
    both = [torch.empty_like(digest) for _ in range(world)]
    dist.all_gather(both, digest)
    tree_same = all(torch.equal(both[0], b) for b in both) and tree.train_timesteps.tolist() == [249, 499, 749, 999, 999, 999]
    # a second model through the rooted-broadcast form must land on the same values
    ud.broadcast_weights(tree2, src=0, bucket_bytes=1 << 20)
    forms_agree = all(torch.equal(a, b) for a, b in zip(tree.state_dict().values(), tree2.state_dict().values()))
    # the config entry point's data sharding: the ranks' shards tile the global batch in order
    from unirestore_amd.data import SyntheticImages
    ds = SyntheticImages(resolution=[16, 24], batch_size=5, num_batches=2, degradations=["noise", "haze", "lowlight"])
    glob = [b[0] for b in ds.batches(0, 1)]
    mine = [b[0] for b in ds.batches(rank, world)]
    szs = [ud.shard_range(5, r, world)[1] - ud.shard_range(5, r, world)[0] for r in range(world)]
    data_ok = all(torch.equal(ud.all_gather_images(m, szs), gl) for m, gl in zip(mine, glob))
    full = torch.arange(7 * 3 * 2 * 2, dtype=torch.float32).view(7, 3, 2, 2)
    lo, hi = ud.shard_range(7, rank, world)
    sizes = [ud.shard_range(7, r, world)[1] - ud.shard_range(7, r, world)[0] for r in range(world)]
    ragged = ud.all_gather_images(full[lo:hi] * 1.0, sizes)       # ragged shards keep batch order
    even = ud.all_gather_images(full[:6][rank * 3:(rank + 1) * 3] * 1.0)
    print(json.dumps(dict(rank=rank, same=same, moved=moved, ragged_ok=torch.equal(ragged, full), even_ok=torch.equal(even, full[:6]),
                          tree_same=tree_same, tree_bytes=n_bytes, forms_agree=forms_agree, data_ok=data_ok)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
