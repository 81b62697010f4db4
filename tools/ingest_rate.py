"""End-to-end rate of the product ingest path: a 720p raw container on disk (page cache) ->
VideoDataset.frame_batches -> fused classify, for whole frames / sampled rows and the
synchronous / overlapped upload.  Prints frames/s including file read, PCIe and classify."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
import numpy as np
import torch
from cutdet import api, rawvideo
from frameID.data import VideoDataset
from frameID.net import load_default_net

n = int(os.environ.get("CUTDET_INGEST_FRAMES", "2048"))
bs = 256
eng = api.default_engine("cuda:0")
cls = torch.tensor([(i // 300) % 3 for i in range(n)], dtype=torch.uint8, device="cuda")
d = tempfile.mkdtemp(dir=os.environ.get("CUTDET_TMP", "/tmp"))
path = os.path.join(d, "clip.cdraw")
with open(path, "wb") as f:
    import struct
    f.write(rawvideo.MAGIC + struct.pack("<6I", 1, 1280, 720, n, 30, 0) + b"\0" * (rawvideo.HEADER_BYTES - 32))
    for a in range(0, n, 256):
        f.write(eng.synth_frames(cls[a:a + 256], 720, 1280, 7, first_index=a).cpu().numpy().tobytes())
net, _ = load_default_net()
net.eval().to("cuda:0")
want = None
for sampled in (False, True):
    for overlap in (False, True):
        def run():
            ds = VideoDataset(path, resize=256, device="cuda:0")
            outs = []
            for b in ds.frame_batches(bs, sampled=sampled, overlap=overlap):
                outs.append(net.classify_frames(b, chan_order=ds.chan_order, net_h=ds.batch_net_height)[2])
            torch.cuda.synchronize()
            return torch.cat(outs)
        got = run()
        assert got.cpu().tolist() == cls.cpu().tolist()
        t = time.perf_counter(); run(); dt = time.perf_counter() - t
        print("%-14s %-22s %8.0f frames/s" % ("sampled rows" if sampled else "whole frames",
                                              "overlapped upload" if overlap else "synchronous upload", n / dt), flush=True)
os.remove(path)
os.rmdir(d)
