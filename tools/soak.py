"""Soak: thousands of fused forwards in both arithmetic modes; every result must equal the first bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
import torch
from cutdet import api
for dtype in ("f32", "bf16"):
    eng = api.default_engine("cuda:0", dtype)
    for (H0, W0, n) in ((720, 1280, 256), (1080, 1920, 64), (180, 320, 100)):
        frames = eng.synth_frames(torch.tensor([i % 3 for i in range(n)], dtype=torch.uint8, device="cuda"), H0, W0, 9)
        ref = [t.clone() for t in eng.forward_u8(frames)]
        bad = 0
        for it in range(1500):
            out = eng.forward_u8(frames)
            if it % 50 == 49:
                bad += int(not all(torch.equal(a, b) for a, b in zip(ref, out)))
        print(dtype, f"{W0}x{H0} n={n}: 1500 launches, mismatching checks: {bad}")
        assert bad == 0
print("soak ok")
