"""Diagnostic: per-wave time stamps of the fused block 1+2 kernel (split bf16).
Needs the library variant built with -DCD_DIAG -DCD_DIAG12:  make -C cut-detection_amd/csrc variant NAME=diag12 FLAGS="-DCD_DIAG -DCD_DIAG12"."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
os.environ["CUTDET_LIB"] = os.path.join(ROOT, "cut-detection_amd", "lib", "variants", "libcutdet_diag12.so")
os.environ["CUTDET_RANGE_ONE_STREAM"] = "1"
os.environ["CUTDET_FUSED12"] = "1"
import torch
from cutdet import api, _lib
eng = api.default_engine("cuda:0", "bf16_split")
n, bs = 1024, 512
frames = eng.synth_frames(torch.zeros(n, dtype=torch.uint8, device="cuda"), 180, 320)     # already 256 x 144: cheap resize
outs = (torch.empty((n, 3), device="cuda"), torch.empty((n,), device="cuda"), torch.empty((n,), dtype=torch.int64, device="cuda"))
for _ in range(5):
    eng.forward_u8_range(frames, bs, outs)
torch.cuda.synchronize()
nwg = 256
buf = np.zeros(nwg * 48, dtype=np.uint64)
off = bs * 18 * 30 * 48 * 4
_lib.lib.cd_debug_peek_act2.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
assert _lib.lib.cd_debug_peek_act2(eng._h, off, buf.nbytes, buf.ctypes.data_as(C.c_void_p)) == 0
s = buf.reshape(nwg, 2, 2, 12).astype(np.int64) / 100.0        # [wg][producer, consumer][tile 1, 2][stamp] in us
P, Cn = s[:, 0], s[:, 1]
d = lambda a, i, j: (a[:, :, j] - a[:, :, i]).mean()
print("producer: wait cells %.2f | produce0 %.2f  barrier %.2f | produce1 %.2f  barrier %.2f | produce2 %.2f  copy issue %.2f  barrier %.2f  (us)" %
      (d(P, 0, 1), d(P, 1, 2), d(P, 2, 3), d(P, 3, 4), d(P, 4, 5), d(P, 5, 6), d(P, 6, 7), d(P, 7, 8)))
print("consumer: consume0 %.2f  barrier %.2f | consume1 %.2f  barrier %.2f | consume2 %.2f  epilogue %.2f  barrier %.2f  (us)" %
      (d(Cn, 0, 1), d(Cn, 1, 2), d(Cn, 2, 3), d(Cn, 3, 4), d(Cn, 4, 5), d(Cn, 5, 6), d(Cn, 6, 7)))
print("tile period: producer %.2f us, consumer %.2f us   (2592 MFMAs of 16 cycles = 17.3 us of one SIMD at 2.4 GHz; two waves share a SIMD)" %
      ((P[:, 1, 0] - P[:, 0, 0]).mean(), (Cn[:, 1, 0] - Cn[:, 0, 0]).mean()))
