"""PCIe-inclusive rate: frames start in pinned host memory, each batch is uploaded and classified.
Whole 720p frames vs only the rows the resize samples (cutdet.engine.sampled_source_rows)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
import torch
from cutdet import api
from cutdet.engine import sampled_source_rows
eng = api.default_engine("cuda:0")
n, bs = 2048, 256
dev_frames = eng.synth_frames(torch.tensor([i % 3 for i in range(n)], dtype=torch.uint8, device="cuda"), 720, 1280, 3)
rows = torch.tensor(sampled_source_rows(720, 1280), device="cuda")
variants = {"whole frames (2.76 MB/frame)": (dev_frames.cpu().pin_memory(), None),
            "sampled rows (553 KB/frame)": (dev_frames[:, rows].contiguous().cpu().pin_memory(), 144)}
ref = eng.forward_u8(dev_frames[:bs])[0].clone()
copy_stream = torch.cuda.Stream()
for name, (host, net_h) in variants.items():
    for overlap in (False, True):
        def run():
            nxt = None
            outs = []
            for a in range(0, n, bs):
                if overlap:
                    if nxt is None:
                        with torch.cuda.stream(copy_stream):
                            nxt = host[a:a + bs].cuda(non_blocking=True)
                    torch.cuda.current_stream().wait_stream(copy_stream)
                    cur = nxt
                    if a + bs < n:
                        with torch.cuda.stream(copy_stream):
                            nxt = host[a + bs:a + 2 * bs].cuda(non_blocking=True)
                else:
                    cur = host[a:a + bs].cuda(non_blocking=True)
                outs.append(eng.forward_u8(cur, net_h=net_h)[0])
                cur.record_stream(torch.cuda.current_stream())
            return outs
        out = run(); torch.cuda.synchronize()
        assert torch.equal(out[0], ref)
        t = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("%-32s %-28s %8.0f frames/s  (%.1f GB/s over PCIe)" % (name, "upload overlapped" if overlap else "upload then classify", n / dt, host.numel() / dt / 1e9))
