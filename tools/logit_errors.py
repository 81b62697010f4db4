"""Largest |logit - reference logit| of every arithmetic on the golden clips (tests/golden/net_golden.npz: logits the
reference itself produced).  usage: python tools/logit_errors.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cutdet import api  # noqa: E402
from cutdet.engine import Engine  # noqa: E402
from oracle import synth  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "net_golden.npz"))
meta = json.load(open(os.path.join(G, "net_golden_meta.json")))
dev = torch.device("cuda:0")
engines = [("f32 (default: block 1 three-term bf16)", Engine(api.default_blob(), dev, "f32"), None),
           ("f32, block 1 on fp32 MFMA", Engine(api.default_blob(), dev, "f32"), "f32_block1_fp32_mfma"),
           ("bf16_split", Engine(api.default_blob(), dev, "bf16_split"), None)]
for name, eng, opt in engines:
    if opt:
        eng.set_option(opt, 1)
    errs = []
    for case in ("720p", "1080p", "180p", "4x3"):
        m = meta[case]
        frames = torch.from_numpy(synth.clip(1234, m["classes"], m["height"], m["width"])).to(dev)
        y = eng.forward_u8(frames)[0].cpu().numpy()
        errs.append(f"{case} {np.abs(y - g[f'logits_{case}']).max():.2e}")
    print(f"{name:42s} " + "  ".join(errs))
