import os, sys, time
sys.path.insert(0, "/root/repo/cut-detection_amd")
import torch
from cutdet import api, synth as csynth
from cutdet.pipeline import ClipClassifier
eng = api.default_engine("cuda:0")
n = 10000
runs = csynth.run_table(n, 1234); cl = csynth.classes_from_runs(runs)
clip = ClipClassifier(eng, n)
# fake class stream straight into the buffers
clip.cls.copy_(torch.tensor(cl, dtype=torch.int64)); clip.maxv.fill_(5.0)
for _ in range(3): clip.segment(n, 100, 10)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    t = time.perf_counter(); r = clip.segment(n, 100, 10); ts.append(time.perf_counter() - t)
print("clip tail (rle + D2H + glue + csv): median %.1f us, min %.1f us" % (sorted(ts)[10] * 1e6, min(ts) * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): clip.segment(n, 100, 10)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
