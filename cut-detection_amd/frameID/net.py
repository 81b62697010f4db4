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
        self.dtype = dtype            # "f32" (the reference's arithmetic), "bf16_split" or "bf16" (see cutdet.Engine)
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

    # --- training-side reuse ------------------------------------------------------------------
    def evaluate(self, batches, chan_order=0):
        """The validation pass of the reference's training script
        (training_scripts/supervised_training.py:177-215) on the fused forward: `batches` yields
        `(x, labels)` pairs or the reference loader's `{"x": ..., "y": ...}` dicts, x being float
        frames [B,3,H,256] or decoded u8 frames [B,H0,W0,3], labels int64 class ids.  Counting and
        the cross-entropy run on the GPU (cd_eval_accumulate).  Returns a dict: `correct`, `total`
        (per class, ints), `accuracy` (per class, correct/total as the script logs it; nan for a
        class without frames), `loss` (mean cross-entropy per frame), `frames`."""
        val = Validation(self)
        for item in batches:
            x, y = (item["x"], item["y"]) if isinstance(item, dict) else item
            val.update(x, y, chan_order=chan_order)
        return val.result()

    def torchscript(self, example=None):
        """What training_scripts/make_torchscript_model.py:25-34 does with the reference's net:
        `torch.jit.trace` on a random [1,3,144,256] input; the returned module can be `.save()`d.  The
        forward is one registered operator (cutdet.torchscript), traced == eager bit for bit.  Only the
        bundled weights can be exported this way (the operator finds them by device, a saved file
        cannot carry a library handle)."""
        from cutdet import api, torchscript
        if bytes(self._blob) != api.default_blob():
            raise NotImplementedError("only the bundled prod_net can be traced; other weights travel as the "
                                      "three model files (save_reference_files)")
        return torchscript.trace(self.dtype, self.engine.device, example)[0]

    def state_dicts(self):
        """(conv_state_dict, linear_state_dict) of torch tensors with the reference's tensor names
        (frameID/net.py FrameConvNet / FrameLinearNet), BatchNorm counters included (as zero: the
        weight blob does not keep them and inference never reads them)."""
        conv, lin = {}, {}
        for name, arr in weights.unpack_blob(self._blob).items():
            (conv if name.startswith("conv_layers.") else lin)[name] = torch.from_numpy(arr.copy())
            if name.endswith("bn.running_var"):
                (conv if name.startswith("conv_layers.") else lin)[name.replace("running_var", "num_batches_tracked")] = \
                    torch.tensor(0, dtype=torch.int64)
        return conv, lin

    def save_reference_files(self, param_file, conv_file, linear_file):
        """Writes the three files `load_and_glue_nets` (here and in the reference, net.py:193-217)
        reads: the JSON hyper-parameters and the two state dicts, as
        supervised_training.py:224-245 saves them."""
        conv, lin = self.state_dicts()
        torch.save(conv, conv_file)
        torch.save(lin, linear_file)
        with open(param_file, "w") as f:
            json.dump(self.params, f)


class Validation:
    """Per-class correct / total counts and the summed cross-entropy of labelled batches, kept on
    the GPU until `result()` (reference training_scripts/supervised_training.py:177-215)."""

    def __init__(self, net, n_classes=3):
        self.net = net
        dev = net.engine.device
        self.correct = torch.zeros((n_classes,), dtype=torch.int64, device=dev)
        self.total = torch.zeros((n_classes,), dtype=torch.int64, device=dev)
        self.loss_sum = torch.zeros((1,), dtype=torch.float64, device=dev)

    def update(self, x, labels, chan_order=0):
        eng = self.net.engine
        x = x.to(eng.device)
        if x.dtype == torch.uint8:
            logits = eng.forward_u8(x, chan_order=chan_order, want_max=False)
        else:
            logits = eng.forward(x)
        labels = labels.to(device=eng.device, dtype=torch.int64).reshape(-1)      # the script squeezes [B,1] -> [B]
        eng.eval_accumulate(logits, labels, self.correct, self.total, self.loss_sum)
        return logits

    def result(self):
        correct, total = self.correct.cpu().tolist(), self.total.cpu().tolist()
        frames = sum(total)
        return {"correct": correct, "total": total,
                "accuracy": [c / t if t else float("nan") for c, t in zip(correct, total)],
                "loss": float(self.loss_sum.item()) / frames if frames else float("nan"), "frames": frames}


# Function to slightly simplify loading networks and making a single callable.
def load_and_glue_nets(param_file, conv_file, linear_file, dtype="f32"):
    """Build the classifier from the reference's three files (JSON hyper-parameters and the
    two state-dict `.pt` files); returns `(net, model_params)`."""
    blob, model_params = weights.pack_reference_files(param_file, conv_file, linear_file)
    return ProdNet(blob, model_params, dtype), model_params


# Load a specific network that's included in the module.
def load_default_net(dtype="f32"):
    """The bundled production classifier (the same weights as the reference's prod_net).  `dtype` is
    an addition to the reference's signature: "bf16_split" and "bf16" select the bf16 matrix-core
    arithmetics (see cutdet.Engine; only "f32" and "bf16_split" keep the logits within 1e-3)."""
    with open(api.DEFAULT_MANIFEST, "r") as f:
        model_params = json.load(f)["model_params"]
    return ProdNet(api.default_blob(), model_params, dtype), model_params
