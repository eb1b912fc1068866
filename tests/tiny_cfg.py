"""Reduced-size configuration shared by the parity tests (same graph, small channels)."""
import torch

TINY = dict(
    unet_cfg=dict(in_channels=4, out_channels=4, block_out_channels=(64, 128, 128, 128), heads=(1, 2, 2, 2),
                  cross_dim=64, groups=32, layers=2),
    vae_cfg=dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(32, 64, 128, 128), groups=32),
    controller_cfg=dict(in_channels=4, model_channels=64, out_channels=64, num_res_blocks=2, dropout=0,
                        channel_mult=(1, 1, 2, 2), downsample_type="conv", num_heads=1,
                        down_block_types=("AttnDownBlock2D",) * 3 + ("DownBlock2D",), mid_block_type="UNetMidBlock2D"),
    fr_depths=(1, 1, 2),
)


def model_kwargs(steps=2):
    return dict(frenc=dict(type="CFRM"), cnet=dict(type="scedit", num_inference_steps=steps),
                tedit=dict(type="TFA", prompt_len=1, task=["ir", "cls", "seg"]))


def randomise_(model, seed=0, zero_init_std=0.05):
    """Re-randomise zero-initialised parameters (SURVEY.md §4 'zero-init trap') deterministically."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            if float(p.abs().sum()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * zero_init_std)
        for name, b in model.named_buffers():
            if name.endswith("null_embeds"):
                b.copy_(torch.randn(b.shape, generator=g))
    return model
