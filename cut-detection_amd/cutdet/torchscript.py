"""The fused forward as a torch operator, so that TorchScript consumers can hold the classifier.

The reference exports its classifier with `torch.jit.trace(net, torch.randn(1, 3, 144, 256))` and
`traced.save(...)` (training_scripts/make_torchscript_model.py:25-34).  Here the network is a handle
on the native library, not a graph of torch modules, so there is nothing for the tracer to record -
unless the forward pass is itself ONE operator.  This module registers it as
`cutdet::prod_net_forward(Tensor x, str dtype) -> Tensor` (torch.library) and wraps it in an
nn.Module; tracing that module records a single call of the operator, and the traced module can be
saved and loaded like the reference's.

What a consumer of the saved file needs: this module imported (the operator is registered from
Python, the file only names it) and an MI355X; the weights are the bundled prod_net's, looked up by
the device of the input.  There is no CPU kernel behind the operator.
"""
import torch

from . import api


@torch.library.custom_op("cutdet::prod_net_forward", mutates_args=())
def prod_net_forward(x: torch.Tensor, dtype: str) -> torch.Tensor:
    """x: float32 [B,3,H,256] RGB in [0,1] on an MI355X -> logits [B,3] (reference net.py:122-133,180-186)."""
    if x.device.type != "cuda":
        raise RuntimeError("cutdet::prod_net_forward runs on an MI355X only (no CPU kernel)")
    return api.default_engine(x.device, dtype).forward(x.to(torch.float32).contiguous())


@prod_net_forward.register_fake
def _(x, dtype):
    return x.new_empty((x.shape[0], 3), dtype=torch.float32)


class ProdNetModule(torch.nn.Module):
    """nn.Module face of the bundled classifier: forward(x) is one call of the operator."""

    def __init__(self, dtype="f32"):
        super().__init__()
        self.dtype = dtype

    def forward(self, x):
        return torch.ops.cutdet.prod_net_forward(x, self.dtype)


def trace(dtype="f32", device="cuda:0", example=None):
    """The reference's export step (make_torchscript_model.py:25-27) on the fused forward: returns
    (traced module, eager module)."""
    module = ProdNetModule(dtype).eval()
    if example is None:
        example = torch.randn(1, 3, 144, 256, device=device)
    return torch.jit.trace(module, example), module
