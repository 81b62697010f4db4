"""Diagnostic: per-wave time stamps of the block-3 + head kernel (needs a -DCD_DIAG -DCD_DIAG3 library build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
os.environ["CUTDET_LIB"] = os.path.join(ROOT, "cut-detection_amd", "lib", "variants", "libcutdet_diag.so")
import torch
from cutdet import api, _lib
eng = api.default_engine("cuda:0")
n = 256
frames = eng.synth_frames(torch.zeros(n, dtype=torch.uint8, device="cuda"), 720, 1280)
for _ in range(40):
    eng.forward_u8(frames)
torch.cuda.synchronize()
buf = np.zeros(n * 4 * 8, dtype=np.uint64)
off = n * 18 * 30 * 48 * 4
_lib.lib.cd_debug_peek_act2.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
assert _lib.lib.cd_debug_peek_act2(eng._h, off, buf.nbytes, buf.ctypes.data_as(C.c_void_p)) == 0
s = buf.reshape(n, 4, 8).astype(np.int64)
t0 = s[:, :, 0].min()
s = (s - t0) / 100.0
for w in range(4):
    print("wave %d stamps (us, mean over workgroups):" % w, " ".join("%.2f" % s[:, w, k].mean() for k in range(8)))
