"""Per-stage HIP-event times of cd_forward_u8_range (stages overlap across streams: each time is the
launch's own duration on its stream, not a share of the wall time).
usage: python tools/range_stages.py [dtype] [height] [width] [batch] [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
import torch  # noqa: E402
from cutdet import api, synth as csynth  # noqa: E402
from cutdet.pipeline import ClipClassifier  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16_split"
h, w, bs, n = (int(v) for v in (sys.argv[2:6] + ["1080", "1920", "512", "4096"][len(sys.argv) - 2:]))
eng = api.default_engine("cuda:0", dtype)
cl = csynth.classes_from_runs(csynth.run_table(n, 1234))
frames = eng.synth_frames(torch.tensor(cl, dtype=torch.uint8, device=eng.device), h, w, 1234)
clip = ClipClassifier(eng, n)
for overlap in (True, False):
    for _ in range(3):
        clip.classify_range(frames, 0, bs, overlap=overlap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        clip.classify_range(frames, 0, bs, overlap=overlap)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    eng.timing_read()
    eng.timing_enable(True)
    clip.classify_range(frames, 0, bs, overlap=overlap)
    torch.cuda.synchronize()
    t = eng.timing_read()
    eng.timing_enable(False)
    print(f"{dtype} {w}x{h} batch {bs} {'range entry' if overlap else 'per-batch calls'}: {n / dt:,.0f} frames/s, "
          f"{dt * 1e6 / (n / bs):.0f} us per batch; per launch: " +
          ", ".join(f"{k} {v[1] * 1e3 / v[0]:.0f} us" for k, v in t.items() if v[0]))
if not os.environ.get("CUTDET_LIB"):            # timing-experiment builds compute wrong values on purpose
    assert clip.cls.cpu().tolist() == cl
