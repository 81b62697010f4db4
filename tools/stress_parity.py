"""Randomised parity sweep of the fused u8 forward against the oracle (both arithmetic modes):
batch sizes that do not divide the XCD grid, padded row strides, cropped views, RGB order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from cutdet import api, _lib
from oracle import net as onet, preprocess as opre
e32 = api.default_engine("cuda:0"); e16 = api.default_engine("cuda:0", "bf16")
w = onet.load_blob()
rng = np.random.default_rng(7)
geoms = [(720, 1280), (1080, 1920), (180, 320), (480, 640), (360, 640), (100, 300), (27 * 4, 256 * 4), (430, 512), (300, 400), (36, 64), (27, 48), (90, 160), (144, 256), (9, 16), (2160, 3840), (54, 96), (600, 800)]
worst32 = worst16 = 0.0
for trial in range(68):
    H0, W0 = geoms[trial % len(geoms)]
    n = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 15, 17, 31]))
    pad = int(rng.choice([0, 0, 3, 16, 37]))
    order = int(rng.integers(0, 2))
    full = torch.from_numpy(rng.integers(0, 256, size=(n, H0 + 2, W0 + pad, 3), dtype=np.uint8)).cuda()
    frames = full[:, 1:H0 + 1, :W0, :]                      # a view: padded rows, offset start, frame stride > H0*row
    x_ref = opre.preprocess_batch(frames.cpu().numpy(), chan_order=order)
    x_gpu = e32.preprocess(frames, chan_order=order)
    assert np.array_equal(x_gpu.cpu().numpy(), x_ref), (trial, H0, W0, n, pad, order)
    ref32 = onet.forward(w, x_ref); ref16 = onet.forward_bf16(w, x_ref)
    l32 = e32.forward_u8(frames, chan_order=order)[0].cpu(); l16 = e16.forward_u8(frames, chan_order=order)[0].cpu()
    d32 = float((l32 - ref32).abs().max()); d16 = float((l16 - ref16).abs().max())
    worst32, worst16 = max(worst32, d32), max(worst16, d16)
    assert d32 <= 1e-3 and d16 <= 5e-3, (trial, H0, W0, n, pad, order, d32, d16)
print("68 trials ok: worst |fp32 - oracle| %.3e, worst |bf16 - bf16 oracle| %.3e" % (worst32, worst16))
