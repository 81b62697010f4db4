"""Weight container for the prod_net classifier.

The native library takes the classifier as ONE flat little-endian blob:

    header   : 8-byte magic b"CDNETW1\\0" + 8 x uint32
               (version, conv_layers, conv_channels, in_channels,
                avg_pool_size, linear_size, linear_output_size, n_floats)
    payload  : float32 tensors, C-contiguous, in state-dict order with the
               BatchNorm `num_batches_tracked` counters dropped:
                 for l in 0..2:  conv.weight [48,Cin,3,3], conv.bias [48],
                                 bn.weight, bn.bias, bn.running_mean, bn.running_var [48]
                 fc0: linear.weight [32,768], linear.bias [32],
                      bn.weight, bn.bias, bn.running_mean, bn.running_var [32]
                 fc1: linear.weight [3,32], linear.bias [3]

The tensor names and shapes are those of the reference's two state dicts
(reference frameID/net.py:193-217 builds the modules that own them).
"""
import json
import struct

import numpy as np

MAGIC = b"CDNETW1\0"
VERSION = 1
HEADER_BYTES = 8 + 8 * 4

# (in_channels, conv_channels, avg_pool, linear_size, n_out) the kernels are built for.
PROD_ARCH = dict(conv_layers=3, conv_channels=48, in_channels=3, avg_pool_size=4,
                 linear_layers=2, linear_size=32, linear_output_size=3)


def tensor_plan(arch=PROD_ARCH):
    """[(name, shape)] in blob order."""
    c = arch["conv_channels"]
    plan = []
    cin = arch["in_channels"]
    for l in range(arch["conv_layers"]):
        p = f"conv_layers.{l}."
        plan.append((p + "conv.weight", (c, cin, 3, 3)))
        plan.append((p + "conv.bias", (c,)))
        for n in ("weight", "bias", "running_mean", "running_var"):
            plan.append((p + "bn." + n, (c,)))
        cin = c
    feat = c * arch["avg_pool_size"] ** 2
    h = arch["linear_size"]
    plan.append(("layers.0.linear.weight", (h, feat)))
    plan.append(("layers.0.linear.bias", (h,)))
    for n in ("weight", "bias", "running_mean", "running_var"):
        plan.append(("layers.0.bn." + n, (h,)))
    plan.append(("layers.1.linear.weight", (arch["linear_output_size"], h)))
    plan.append(("layers.1.linear.bias", (arch["linear_output_size"],)))
    return plan


def check_arch(params):
    """Raise ValueError unless `params` (the reference's model_params JSON dict) is prod_net."""
    for k, v in PROD_ARCH.items():
        if k == "in_channels":
            continue
        if k in params and int(params[k]) != v:
            raise ValueError(
                f"cutdet kernels are built for prod_net ({PROD_ARCH}); got {k}={params[k]}")


def pack_state_dicts(conv_sd, linear_sd, params=None):
    """Pack the two state dicts (name -> array-like) into a blob (bytes)."""
    if params is not None:
        check_arch(params)
    merged = {}
    merged.update(conv_sd)
    merged.update(linear_sd)
    chunks = []
    n = 0
    for name, shape in tensor_plan():
        if name not in merged:
            raise KeyError(f"missing tensor {name}")
        a = np.ascontiguousarray(np.asarray(merged[name], dtype=np.float32))
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {a.shape}, expected {shape}")
        chunks.append(a.astype("<f4").tobytes())
        n += a.size
    a = PROD_ARCH
    head = MAGIC + struct.pack("<8I", VERSION, a["conv_layers"], a["conv_channels"],
                               a["in_channels"], a["avg_pool_size"], a["linear_size"],
                               a["linear_output_size"], n)
    return head + b"".join(chunks)


def unpack_blob(blob):
    """blob bytes -> {name: float32 ndarray}. Validates magic, version, architecture, size."""
    if len(blob) < HEADER_BYTES or blob[:8] != MAGIC:
        raise ValueError("not a cutdet weight blob (bad magic)")
    ver, nl, cc, ic, ap, ls, no, n = struct.unpack("<8I", blob[8:HEADER_BYTES])
    a = PROD_ARCH
    if (ver, nl, cc, ic, ap, ls, no) != (VERSION, a["conv_layers"], a["conv_channels"],
                                         a["in_channels"], a["avg_pool_size"],
                                         a["linear_size"], a["linear_output_size"]):
        raise ValueError("weight blob is not a version-1 prod_net blob")
    if len(blob) != HEADER_BYTES + 4 * n:
        raise ValueError("weight blob truncated")
    flat = np.frombuffer(blob, dtype="<f4", offset=HEADER_BYTES)
    out, pos = {}, 0
    for name, shape in tensor_plan():
        k = int(np.prod(shape))
        out[name] = flat[pos:pos + k].reshape(shape).copy()
        pos += k
    if pos != n:
        raise ValueError("weight blob float count mismatch")
    return out


def load_torch_state_dict(path):
    """Read a reference `.pt` state dict without executing anything from the file."""
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return {k: v.detach().cpu().numpy() for k, v in sd.items() if v.dtype.is_floating_point}


def pack_reference_files(param_file, conv_file, linear_file):
    """(blob bytes, params dict) from the reference's three model files
    (reference frameID/net.py:193-217 reads the same three)."""
    with open(param_file, "r") as f:
        params = json.load(f)
    blob = pack_state_dicts(load_torch_state_dict(conv_file),
                            load_torch_state_dict(linear_file), params)
    return blob, params
