"""How far are the block-1 activations of the two fp32 formulations from exact arithmetic?
Default: frame bytes x three exact bfloat16 weight terms on the bf16 matrix pipe; option f32_block1_fp32_mfma:
v_mfma_f32_16x16x4_f32.  Reference value: conv + bias -> ReLU -> 3x3 max-pool -> BatchNorm in float64 on the
real numbers k/255 (reference frameID/net.py:33-40 on frameID/data.py:227's division, without any rounding).
Needs a diagnostic build (it reads the activation workspace):
    make -C cut-detection_amd/csrc variant NAME=diag FLAGS=-DCD_DIAG
usage: python tools/block1_exactness.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CUTDET_LIB"] = os.path.join(ROOT, "cut-detection_amd", "lib", "variants", "libcutdet_diag.so")
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from cutdet import _lib, api  # noqa: E402
from cutdet.engine import Engine  # noqa: E402
from oracle import net as onet, preprocess, synth  # noqa: E402

H, W, N = 720, 1280, 4
frames = synth.clip(1234, [0, 1, 2, 1], H, W)
w = onet.load_blob()
x8 = preprocess.preprocess_batch(frames)                    # float32 k/255, CHW RGB: only used for its shape / order
k = np.rint(x8.astype(np.float64) * 255.0)                  # the bytes back, exactly
x = torch.from_numpy(k / 255.0)                             # float64 real k/255
with torch.no_grad():
    y = F.conv2d(x, w["c0.w"].double(), w["c0.b"].double(), padding=1)
    y = F.max_pool2d(F.relu(y), kernel_size=3)
    y = F.batch_norm(y, w["c0.mu"].double(), w["c0.var"].double(), w["c0.g"].double(), w["c0.beta"].double(),
                     training=False, eps=onet.BN_EPS)
want = y.permute(0, 2, 3, 1).numpy()                        # [N, h1, 85, 48]
h1, w1 = want.shape[1], want.shape[2]

_lib.lib.cd_debug_peek_act1.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
dev = torch.device("cuda:0")
for name, opt in (("default (bytes x three bfloat16 terms, bf16 matrix pipe)", 0), ("v_mfma_f32_16x16x4_f32", 1)):
    eng = Engine(api.default_blob(), dev, "f32")
    eng.set_option("f32_block1_fp32_mfma", opt)
    eng.forward_u8(torch.from_numpy(frames).to(dev))
    torch.cuda.synchronize()
    buf = np.empty((N, h1 + 2, w1 + 2, 48), dtype=np.float32)
    assert _lib.lib.cd_debug_peek_act1(eng._h, 0, buf.nbytes, buf.ctypes.data_as(C.c_void_p)) == 0
    got = buf[:, 1:-1, 1:-1, :].astype(np.float64)
    err = np.abs(got - want)
    rel = err / np.maximum(np.abs(want), 1e-3)
    print(f"{name:58s} max |err| {err.max():.3e}  mean |err| {err.mean():.3e}  max rel {rel.max():.3e}  "
          f"(activations up to {np.abs(want).max():.2f})")
    eng.close()
