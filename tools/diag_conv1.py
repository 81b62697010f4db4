"""Diagnostic: per-workgroup time stamps of the block-1 kernel (needs the CD_DIAG library build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
os.environ["CUTDET_LIB"] = os.path.join(ROOT, "cut-detection_amd", "lib", "variants", "libcutdet_diag.so")
import torch
from cutdet import api, _lib
eng = api.default_engine("cuda:0")
n = 256
cl = torch.zeros(n, dtype=torch.uint8, device="cuda")
frames = eng.synth_frames(cl, 720, 1280)
for _ in range(3):
    eng.forward_u8(frames)
torch.cuda.synchronize()
nwg = n * 16
buf = np.zeros(nwg * 16, dtype=np.uint64)
off = n * 50 * 87 * 48 * 4
_lib.lib.cd_debug_peek_act1.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
rc = _lib.lib.cd_debug_peek_act1(eng._h, off, buf.nbytes, buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
s = buf.reshape(nwg, 16).astype(np.int64)
t0 = s[:, 0].min()
start, mid, end = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0, (s[:, 2] - t0) / 100.0   # 100 MHz -> us
hw = s[:, 3] & 0xFFFFFFFF
xcc = s[:, 3] >> 32
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 16 + cu
print("kernel span us:", end.max(), " WGs:", nwg, " distinct CU ids:", len(np.unique(cuid)))
print("stage phase us: mean %.2f p50 %.2f p90 %.2f max %.2f" % ((mid - start).mean(), np.median(mid - start), np.percentile(mid - start, 90), (mid - start).max()))
a, b, c = (s[:, 4] - t0) / 100.0, (s[:, 5] - t0) / 100.0, (s[:, 6] - t0) / 100.0
f = (s[:, 7] - t0) / 100.0
u0, u4, u8 = (s[:, 10] - t0) / 100.0, (s[:, 11] - t0) / 100.0, (s[:, 12] - t0) / 100.0
for nm, x in (("start->first gather", u0 - start), ("gathers 0-3", u4 - u0), ("gathers 4-7", u8 - u4), ("gathers 8-11", f - u8), ("gathers issued->before commit", a - f), ("wait for gathers", b - a), ("commit (LUT + LDS writes, wave 0)", c - b), ("barrier wait", mid - c)):
    print("  %-40s mean %.2f p50 %.2f p90 %.2f" % (nm, x.mean(), np.median(x), np.percentile(x, 90)))
print("compute phase us: mean %.2f p50 %.2f p90 %.2f max %.2f" % ((end - mid).mean(), np.median(end - mid), np.percentile(end - mid, 90), (end - mid).max()))
print("WG life us: mean %.2f" % (end - start).mean())
# concurrency: average number of resident WGs per CU over the kernel span
tot = (end - start).sum()
print("avg resident WGs per CU: %.2f (sum life / span / CUs)" % (tot / end.max() / len(np.unique(cuid))))
per = {}
for c, a, b in zip(cuid, start, end):
    per.setdefault(c, []).append((a, b))
cnts = [len(v) for v in per.values()]
print("WGs per CU: min %d max %d" % (min(cnts), max(cnts)))
lastend = np.array([max(b for a, b in v) for v in per.values()])
print("per-CU last end us: min %.1f p50 %.1f max %.1f" % (lastend.min(), np.median(lastend), lastend.max()))
print("start time of WGs: p50 %.1f p90 %.1f max %.1f" % (np.median(start), np.percentile(start, 90), start.max()))
# phase clustering: how many workgroups of a CU are in their staging phase at the same time
hist = np.histogram(start, bins=np.arange(0, end.max() + 5, 5))[0]
print("WG starts per 5 us bin:", " ".join(str(v) for v in hist))
grid = np.arange(0, end.max(), 0.5)
staging = np.zeros((len(per), len(grid)), dtype=np.int32)
resident = np.zeros((len(per), len(grid)), dtype=np.int32)
ids = {c: k for k, c in enumerate(per)}
for c, a, m, b in zip(cuid, start, mid, end):
    k = ids[c]
    staging[k] += (grid >= a) & (grid < m)
    resident[k] += (grid >= a) & (grid < b)
computing = resident - staging
print("time share of a CU with k workgroups in the MFMA phase: " + " ".join("k=%d %.3f" % (k, (computing == k).mean()) for k in range(6)))
print("time share of a CU with k workgroups staging:           " + " ".join("k=%d %.3f" % (k, (staging == k).mean()) for k in range(6)))
