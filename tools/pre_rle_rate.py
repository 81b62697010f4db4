"""HBM-side kernels on their own: stand-alone resize+normalise and the run-length encode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
import torch
from cutdet import api
eng = api.default_engine("cuda:0")
for (H0, W0, n, rows) in ((720, 1280, 256, 144), (1080, 1920, 256, 288), (180, 320, 1024, 180)):
    frames = eng.synth_frames(torch.zeros(n, dtype=torch.uint8, device="cuda"), H0, W0, 1)
    out = eng.preprocess(frames)
    for _ in range(5): eng.preprocess(frames, out=out)
    eng.timing_read(); eng.timing_enable(True, stages=["preprocess"])
    for _ in range(50): eng.preprocess(frames, out=out)
    k, ms = eng.timing_read()["preprocess"]; eng.timing_enable(False)
    us = ms / k * 1e3
    rd, wr = n * rows * W0 * 3, out.numel() * 4
    print("preprocess %4dx%-4d n=%4d: %7.1f us  sampled rows %6.1f MB + floats %6.1f MB -> %5.2f TB/s (%.0f%% of 8)" % (W0, H0, n, us, rd / 1e6, wr / 1e6, (rd + wr) / us / 1e6, (rd + wr) / us / 1e6 / 8 * 100))
n = 1 << 24
cls = (torch.arange(n, device="cuda") // 700 % 3).to(torch.int64); mx = torch.rand(n, device="cuda")
for _ in range(3): eng.rle(cls, mx)
eng.timing_read(); eng.timing_enable(True, stages=["rle"])
for _ in range(10): eng.rle(cls, mx)
k, ms = eng.timing_read()["rle"]; eng.timing_enable(False)
us = ms / k * 1e3
print("rle n=%d frames: %.1f us -> %.2f TB/s of the 12 B/frame stream read once (the scan reads it twice)" % (n, us, n * 12 / us / 1e6))
