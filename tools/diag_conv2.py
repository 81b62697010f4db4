"""Diagnostic: per-wave time stamps of the block-2 kernel (needs the CD_DIAG library build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
# usage: diag_conv2.py [dtype] [variant]   (the variant must be built with -DCD_DIAG -DCD_DIAG_P=<0|1|2> for that dtype)
DTYPE = sys.argv[1] if len(sys.argv) > 1 else "f32"
os.environ["CUTDET_LIB"] = os.path.join(ROOT, "cut-detection_amd", "lib", "variants",
                                        "libcutdet_%s.so" % (sys.argv[2] if len(sys.argv) > 2 else "diag"))
import torch
from cutdet import api, _lib
eng = api.default_engine("cuda:0", DTYPE)
n = 256
frames = eng.synth_frames(torch.zeros(n, dtype=torch.uint8, device="cuda"), 720, 1280)
for _ in range(40):
    eng.forward_u8(frames)
torch.cuda.synchronize()
nwg = 200   # the slack frame holds the stamps of the first 200 waves (tile marks, then per-step marks)
buf = np.zeros(2 * nwg * 32, dtype=np.uint64)
off = n * 18 * 30 * 48 * 4
_lib.lib.cd_debug_peek_act2.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
assert _lib.lib.cd_debug_peek_act2(eng._h, off, buf.nbytes, buf.ctypes.data_as(C.c_void_p)) == 0
steps = buf[nwg * 32:].reshape(nwg, 32).astype(np.int64)
s = buf[:nwg * 32].reshape(nwg, 32).astype(np.int64)
t0 = s[:, 0].min()
s = (s - t0) / 100.0
# layout: 0 start, 1 loop entry, then per tile: unit0, unit1, unit2, epilogue
print("wave start: p50 %.2f max %.2f us; prologue (first copy + first operands): mean %.2f us" % (np.median(s[:, 0]), s[:, 0].max(), (s[:, 1] - s[:, 0]).mean()))
idx = 1
for tile in range(7):
    u = [s[:, idx + 1 + g] - s[:, idx + g] for g in range(3)]
    e = s[:, idx + 4] - s[:, idx + 3]
    print("tile %d: units %.2f %.2f %.2f us  epilogue %.2f us" % (tile, u[0].mean(), u[1].mean(), u[2].mean(), e.mean()))
    idx += 4
print("wave end: p50 %.2f p90 %.2f max %.2f us  (972 MFMAs = 12.96 us at 2.4 GHz)" % (np.median(s[:, 29]), np.percentile(s[:, 29], 90), s[:, 29].max()))
d = np.diff(steps.astype(np.float64) / 100.0, axis=1)
print("fourth tile, per unit: [entry->step0] then step durations (us, mean over waves; 108 MFMAs = 1.45 us at 2.39 GHz):")
for g in range(3):
    o = 10 * g
    print("  unit %d: entry %.2f | " % (g, (steps[:, o] - steps[:, o + 9]).mean() / 100.0) + " ".join("%.2f" % d[:, o + k].mean() for k in range(8)))
