"""Runs the fused u8 forward a few times on one batch (profiling target for PMC passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
import torch
from cutdet import api
eng = api.default_engine("cuda:0")
n = 256
frames = eng.synth_frames(torch.zeros(n, dtype=torch.uint8, device="cuda"), 720, 1280)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    eng.forward_u8(frames)
torch.cuda.synchronize()
