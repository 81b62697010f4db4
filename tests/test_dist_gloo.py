"""The N > 1 stitch on CPU: two gloo ranks gather their (class, max) streams; the gathered
stream must equal the single-process stream and segment identically."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _streams(n, seed=5):
    rs = np.random.RandomState(seed)
    runs, total = [], 0
    while total < n:
        k = int(rs.choice([3, 12, 150, 400]))
        runs.append(k)
        total += k
    cls = np.concatenate([np.full(r, i % 3, np.int64) for i, r in enumerate(runs)])[:n]
    mx = rs.uniform(1, 9, size=n).astype(np.float32)
    return cls, mx


def _worker(rank, world, port, n, split, out_dir):
    for p in (ROOT, os.path.join(ROOT, "cut-detection_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cutdet.pipeline import gather_streams
    cls, mx = _streams(n)
    lo, hi = (0, split) if rank == 0 else (split, n)          # unequal shards on purpose
    g_cls, g_mx = gather_streams(torch.from_numpy(cls[lo:hi]), torch.from_numpy(mx[lo:hi]))
    np.save(os.path.join(out_dir, f"cls{rank}.npy"), g_cls.numpy())
    np.save(os.path.join(out_dir, f"mx{rank}.npy"), g_mx.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reproduces_the_single_stream(tmp_path):
    from oracle import segmentation
    n, split = 5000, 1873
    mp.spawn(_worker, args=(2, _free_port(), n, split, str(tmp_path)), nprocs=2, join=True)
    cls, mx = _streams(n)
    for r in range(2):
        assert (np.load(tmp_path / f"cls{r}.npy") == cls).all()
        assert np.load(tmp_path / f"mx{r}.npy").tobytes() == mx.tobytes()
    # segmenting the gathered stream == segmenting the original
    scores = np.full((n, 3), -5.0, np.float32)
    scores[np.arange(n), cls] = mx
    g = np.full((n, 3), -5.0, np.float32)
    g[np.arange(n), np.load(tmp_path / "cls0.npy")] = np.load(tmp_path / "mx0.npy")
    a = segmentation.segment_clip(scores)
    b = segmentation.segment_clip(g)
    assert a[3] == b[3] and a[2]["start_frames"] == b[2]["start_frames"]


def test_gather_is_identity_without_a_process_group():
    for p in (ROOT, os.path.join(ROOT, "cut-detection_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from cutdet.pipeline import gather_streams
    c, m = torch.arange(5), torch.rand(5)
    gc, gm = gather_streams(c, m)
    assert gc is c and gm is m
