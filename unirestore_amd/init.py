"""Seeded random weights of the reference architecture (no checkpoint is reachable offline): what `bench.py`, the CLI and
the full-size tests run the path on.  N(0, 0.7^2/fan_in) weights, unit norm gains, zero biases, small non-zero values for
the parameters the reference initialises to zero (zero-convs of the Controller, NAFBlock beta / gamma, task prompts) so
that no branch of the graph is numerically dead."""
import torch


def init_random_(model, seed: int, device):
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            leaf = name.rsplit(".", 1)[-1]
            if p.dim() >= 2 and leaf == "weight":
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g, device=device) * (0.7 / fan_in ** 0.5))
            elif leaf == "weight":           # norm gains
                p.fill_(1.0)
            elif leaf == "bias":
                p.zero_()
            elif leaf in ("beta", "gamma"):
                p.fill_(0.1)
            else:                            # task prompts
                p.copy_(torch.randn(p.shape, generator=g, device=device) * 0.02)
        for name, b in model.named_buffers():
            if name.endswith("null_embeds") and float(b.abs().sum()) == 0.0:
                b.copy_(torch.randn(b.shape, generator=g, device=device))
    return model
