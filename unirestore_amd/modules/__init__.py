"""Reference-compatible operator classes (src/modules/diffuie/*) backed by the HIP kernels."""
from .adapters import SPADE, AdaNAFV2, CSCEAdapter, NAFBlock, TaskEditorV1c, TaskFeatureAdapter, cfrm_blocks
from .model import Controller, ControlledUNet, DiffUIE, SkipConnectedAutoEncoder, resize_pad_plan, stablesr_config
from .nn import AutoencoderKL, UNet2DConditionModel

__all__ = ["SPADE", "AdaNAFV2", "CSCEAdapter", "NAFBlock", "TaskEditorV1c", "TaskFeatureAdapter", "cfrm_blocks", "Controller",
           "ControlledUNet", "DiffUIE", "SkipConnectedAutoEncoder", "resize_pad_plan", "stablesr_config",
           "AutoencoderKL", "UNet2DConditionModel"]
