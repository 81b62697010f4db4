"""Does replaying captured HIP graphs (one per batch of the clip) beat issuing the three launches per batch?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
import torch
from cutdet import api
eng = api.default_engine("cuda:0")
n, bs = 10240, 256
clip = eng.synth_frames(torch.tensor([i % 3 for i in range(n)], dtype=torch.uint8, device="cuda"), 720, 1280, 1)
logits = torch.empty((n, 3), device="cuda"); mx = torch.empty((n,), device="cuda"); am = torch.empty((n,), dtype=torch.int64, device="cuda")
def direct():
    for a in range(0, n, bs):
        eng.forward_u8(clip[a:a + bs], out=(logits[a:a + bs], mx[a:a + bs], am[a:a + bs]))
for _ in range(2): direct()
torch.cuda.synchronize()
graphs = []
for a in range(0, n, bs):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.forward_u8(clip[a:a + bs], out=(logits[a:a + bs], mx[a:a + bs], am[a:a + bs]))
    graphs.append(g)
def replay():
    for g in graphs: g.replay()
for name, f in (("direct launches", direct), ("graph replay", replay), ("direct launches", direct), ("graph replay", replay)):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print("%-16s %.1f us per 256-frame batch, %.0f frames/s" % (name, dt / (n / bs) * 1e6, n / dt))
