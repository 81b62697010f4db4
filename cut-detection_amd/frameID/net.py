"""The prod_net classifier behind the reference's loader functions.

`load_default_net()` / `load_and_glue_nets()` keep the names, arguments and return value
`(net, model_params)` of reference frameID/net.py:193-233.  `net` is not an nn.Module: it is
a handle on the fused HIP forward pass (cutdet.Engine) that supports what the reference's
driver uses (segment_video.py:31-45): `net.eval()`, `net.to(device)`, `net(batch) -> [B,3]`
fp32 logits on the batch's device.  The network runs on an MI355X only; calling it while it
is on "cpu" raises instead of falling back.
"""
import json
import os

import torch

from cutdet import api, weights
from cutdet.engine import Engine

package_directory = os.path.dirname(os.path.abspath(__file__))


class ProdNet:
    """Inference-only prod_net (3 x [conv3x3, ReLU, maxpool3, BN] + avgpool4 + 2 FC)."""

    training = False

    def __init__(self, blob, params, dtype="f32"):
        self._blob = blob
        self.params = dict(params)
        self.dtype = dtype            # "f32" (the reference's arithmetic) or "bf16" (see cutdet.Engine)
        self._device = torch.device("cpu")
        self._engine = None

    # --- the nn.Module-ish surface the driver touches ------------------------------------
    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise RuntimeError("ProdNet is inference-only (BatchNorm uses its running statistics)")
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda":
            idx = device.index if device.index is not None else torch.cuda.current_device()
            device = torch.device("cuda", idx)
            if self._engine is None or self._engine.device != device:
                self._engine = Engine(self._blob, device, self.dtype)
        self._device = device
        return self

    def cuda(self, index=0):
        return self.to(torch.device("cuda", index))

    @property
    def engine(self):
        if self._device.type != "cuda" or self._engine is None:
            raise RuntimeError(
                "ProdNet has no CPU implementation: move it to an MI355X with net.to('cuda:0') "
                "(the reference's --cpu mode is not available in this build)")
        return self._engine

    def __call__(self, batch):
        """[B,3,H,256] fp32 RGB in [0,1] on the GPU -> logits [B,3] (reference net.py:122-133,180-186)."""
        return self.engine.forward(batch)

    forward = __call__

    def classify_frames(self, frames_u8, chan_order=0, net_h=None):
        """Decoded u8 frames [B,H0,W0,3] on the GPU -> (logits, max, argmax); resize fused in.
        `net_h`: see cutdet.Engine.forward_u8 (frames already reduced to their sampled rows)."""
        return self.engine.forward_u8(frames_u8, chan_order=chan_order, net_h=net_h)


# Function to slightly simplify loading networks and making a single callable.
def load_and_glue_nets(param_file, conv_file, linear_file, dtype="f32"):
    """Build the classifier from the reference's three files (JSON hyper-parameters and the
    two state-dict `.pt` files); returns `(net, model_params)`."""
    blob, model_params = weights.pack_reference_files(param_file, conv_file, linear_file)
    return ProdNet(blob, model_params, dtype), model_params


# Load a specific network that's included in the module.
def load_default_net(dtype="f32"):
    """The bundled production classifier (the same weights as the reference's prod_net).  `dtype` is
    an addition to the reference's signature: "bf16" selects the bf16 matrix-core arithmetic."""
    with open(api.DEFAULT_MANIFEST, "r") as f:
        model_params = json.load(f)["model_params"]
    return ProdNet(api.default_blob(), model_params, dtype), model_params
