"""Worker for tests/test_dist_gpu.py: the N > 1 code path on ONE GPU (world size 1 over RCCL) - process-group init, scatter +
all-gather weight broadcast, hipGraph capture beside the RCCL watchdog thread, output all-gather after the replay - and the
result must equal the non-distributed run of the same weights / inputs bit for bit.  Prints a JSON verdict."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist


def main():
    from tiny_cfg import TINY, model_kwargs, randomise_
    import unirestore_amd.modules as M
    from unirestore_amd import dist as ud
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 3, 64, 64, generator=g)
    noise = (torch.randn(2, 4, 64, 64, generator=g), torch.randn(2, 4, 64, 64, generator=g))
    # (1) plain single-process run
    a = randomise_(M.DiffUIE(**model_kwargs(2), **TINY).eval(), 11)
    ref = a(img, "ir", noise=noise).clone()
    ref2 = a(img, "ir", noise=noise).clone()          # graph replay
    # (2) the distributed path at world size 1: a second model (its own random init replaced by `a`'s weights) goes through the
    #     scatter + all-gather broadcast - at world size 1 the identity, but every collective, bucket and copy-back runs
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    b = M.DiffUIE(**model_kwargs(2), **TINY).eval()
    b.load_state_dict(a.state_dict())
    b = b.to(dev)
    moved = ud.broadcast_weights_sharded(b, src=0, bucket_bytes=1 << 20)
    dist.barrier()
    b.refresh()
    out = b(img, "ir", noise=noise)                   # captures the graph while the RCCL watchdog thread is alive
    gathered = ud.all_gather_images(out, [2])
    out2 = b(img, "ir", noise=noise)                  # replay
    gathered2 = torch.empty(2, *out2.shape[1:], device=dev)
    dist.all_gather_into_tensor(gathered2, out2.contiguous())
    torch.cuda.synchronize()
    verdict = dict(moved=int(moved), same_as_plain=bool(torch.equal(gathered.cpu(), ref.cpu())), replay_same=bool(torch.equal(gathered2.cpu(), ref2.cpu())),
                   plain_replay_same=bool(torch.equal(ref.cpu(), ref2.cpu())), finite=bool(torch.isfinite(gathered).all()))
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(verdict))


if __name__ == "__main__":
    main()
