"""A/B of library builds: per-stage kernel times (HIP events), variants interleaved and repeated so that
box drift and run-to-run noise (about 2 % on a single bench run) average out.

usage: python tools/ab_stage.py [--dtype f32] [--reps 3] lib_a.so lib_b.so ...
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%r, "cut-detection_amd"))
import torch
from cutdet import api
eng = api.default_engine("cuda:0", %r)
# a clip far larger than the caches (like the bench): every batch's source rows come from HBM
n = 10240
clip = eng.synth_frames(torch.tensor([i %% 3 for i in range(n)], dtype=torch.uint8, device="cuda"), 720, 1280, 1)
batches = [clip[a:a + 256] for a in range(0, n, 256)]
for i in range(40): eng.forward_u8(batches[i %% 40])
torch.cuda.synchronize()
eng.timing_enable(True)
for i in range(160): eng.forward_u8(batches[i %% 40])
t = eng.timing_read()
print(json.dumps({k: v[1] / v[0] * 1e3 for k, v in t.items() if v[0]}))
'''

def main():
    args = sys.argv[1:]
    dtype, reps = "f32", 3
    while args and args[0].startswith("--"):
        if args[0] == "--dtype": dtype = args[1]
        if args[0] == "--reps": reps = int(args[1])
        args = args[2:]
    res = {lib: [] for lib in args}
    for _ in range(reps):
        for lib in args:
            env = dict(os.environ, CUTDET_LIB=os.path.abspath(lib))
            out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, dtype)], env=env, capture_output=True, text=True, timeout=300)
            if out.returncode:
                print(out.stderr[-2000:]); sys.exit(1)
            res[lib].append(json.loads(out.stdout.strip().splitlines()[-1]))
    for lib, runs in res.items():
        keys = runs[0].keys()
        med = {k: sorted(r[k] for r in runs)[len(runs) // 2] for k in keys}
        print("%-48s" % os.path.basename(lib), " ".join("%s %.1f" % (k, med[k]) for k in keys), " sum %.1f us" % sum(med.values()),
              " | conv1 runs:", " ".join("%.1f" % r["conv1"] for r in runs))

main()
