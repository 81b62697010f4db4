"""The N > 1 path on CPU: gloo ranks gather their (class, max) streams with the job's one
all-gather; the gathered stream must equal the single-process stream and segment identically.
Also: bench.py starts its own ranks when no launcher did."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
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


def _scores(cls, mx):
    s = np.full((cls.shape[0], 3), -5.0, np.float32)
    s[np.arange(cls.shape[0]), cls] = mx
    return s


def _worker(rank, world, port, n, out_dir):
    for p in (ROOT, os.path.join(ROOT, "cut-detection_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cutdet import pipeline
    from oracle import segmentation
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    cls, mx = _streams(n)
    lo, hi = pipeline.shard_range(n, rank, world)     # ceil(n / world) per rank: unequal when n % world != 0
    g_cls, g_mx = pipeline.gather_streams(torch.from_numpy(cls[lo:hi]), torch.from_numpy(mx[lo:hi]), total=n)
    dist.all_gather_into_tensor = real
    assert len(calls) == 1, "the stitch must be ONE collective"
    np.save(os.path.join(out_dir, f"cls{rank}.npy"), g_cls.numpy())
    np.save(os.path.join(out_dir, f"mx{rank}.npy"), g_mx.numpy())
    # what ClipClassifier.segment does with the gathered streams, with the GPU run-length pass
    # replaced by the oracle's: the product's host repair (cd_glue_segments) and CSV rendering
    te = segmentation.segments_from_scores(_scores(g_cls.numpy(), g_mx.numpy()))
    initial = {k: np.asarray(te[k], dtype=np.float32 if k == "score_means" else np.int64) for k in pipeline.TABLE_KEYS}
    out = pipeline.finish_table(initial, 100, 10)
    with open(os.path.join(out_dir, f"csv{rank}.bin"), "wb") as f:
        f.write(out["csv"])
    # ONE rank holds a count the split does not give it: the collective still completes (fixed-size
    # payloads) and EVERY rank raises from the gathered headers - nobody is left inside the all-gather
    cut = 1 if rank == world - 1 else 0
    try:
        pipeline.gather_streams(torch.from_numpy(cls[lo:hi - cut]), torch.from_numpy(mx[lo:hi - cut]), total=n)
        raise AssertionError("wrong shard size accepted")
    except ValueError as e:
        assert f"rank {world - 1} holds" in str(e)
    # without `total` the counts cannot be known: refused before any collective
    try:
        pipeline.gather_streams(torch.from_numpy(cls[lo:hi]), torch.from_numpy(mx[lo:hi]))
        raise AssertionError("missing total accepted")
    except ValueError:
        pass
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 5001), (3, 4000)])
def test_ranks_gather_and_segment_like_one_process(tmp_path, world, n):
    from oracle import segmentation
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    cls, mx = _streams(n)
    want = segmentation.segment_clip(_scores(cls, mx))
    for r in range(world):
        assert (np.load(tmp_path / f"cls{r}.npy") == cls).all()
        assert np.load(tmp_path / f"mx{r}.npy").tobytes() == mx.tobytes()
        assert (tmp_path / f"csv{r}.bin").read_bytes() == want[3]


def test_shard_ranges_are_contiguous_and_cover():
    for p in (ROOT, os.path.join(ROOT, "cut-detection_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from cutdet.pipeline import shard_range
    for total, world in [(200000, 8), (10, 4), (3, 8), (0, 2), (10001, 3)]:
        edges = [shard_range(total, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        assert max(hi - lo for lo, hi in edges) == -(-total // world)


def test_gather_is_identity_without_a_process_group():
    for p in (ROOT, os.path.join(ROOT, "cut-detection_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from cutdet.pipeline import gather_streams
    c, m = torch.arange(5), torch.rand(5)
    gc, gm = gather_streams(c, m)
    assert gc is c and gm is m


def _bench(args, **env):
    e = dict(os.environ, CUTDET_BENCH_REHEARSE="plumbing", **env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True,
                          text=True, timeout=300)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: two child ranks rendezvous, gather, rank 0 prints
    one JSON line (plumbing rehearsal: no GPU work on this machine)."""
    r = _bench(["--gpus", "2", "--clip-frames", "3000"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["gathered_frames"] == 6000 and out["gather_ok"]
    # the rendezvous of the no-process-group leg (file barrier + result exchange) saw both ranks
    assert out["file_barrier_ranks"] == 2 and out["file_barrier_frames"] == 6000


def test_bench_strong_split_and_failed_rank():
    r = _bench(["--gpus", "3", "--total-frames", "10001"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["gathered_frames"] == 10001 and out["gather_ok"] and out["n_ranks_seen"] == 3
    # a rank that dies takes the job down with its exit code; the surviving rank is stopped
    r = _bench(["--gpus", "2", "--clip-frames", "2000"], CUTDET_BENCH_FAIL_RANK="1")
    assert r.returncode == 7
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_file_barrier_lines_up_processes(tmp_path):
    """cutdet.launch.FileBarrier: ranks without a process group meet through files."""
    for p in (ROOT, os.path.join(ROOT, "cut-detection_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import threading
    from cutdet.launch import FileBarrier
    world, got = 4, {}

    def rank_body(r):
        fb = FileBarrier(r, world, str(tmp_path), timeout=30)
        fb.wait()
        got[r] = fb.exchange("value", str(r * r))
        fb.wait()

    threads = [threading.Thread(target=rank_body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert all(got[r] == ["0", "1", "4", "9"] for r in range(world))
    os.makedirs(tmp_path / "nobody")
    with pytest.raises(TimeoutError):            # the other rank never arrives
        FileBarrier(0, 2, str(tmp_path / "nobody"), timeout=0.2).wait()


def test_raw_capture_seeks_like_cv2(tmp_path):
    """RawVideoCapture.set(CAP_PROP_POS_FRAMES): a rank opens the clip at its frame range."""
    for p in (ROOT, os.path.join(ROOT, "cut-detection_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from cutdet import rawvideo
    frames = np.arange(7 * 4 * 6 * 3, dtype=np.uint8).reshape(7, 4, 6, 3)
    path = str(tmp_path / "c.cdraw")
    rawvideo.write_raw_video(path, frames)
    cap = rawvideo.RawVideoCapture(path)
    assert cap.set(rawvideo.CAP_PROP_POS_FRAMES, 5) and cap.get(rawvideo.CAP_PROP_POS_FRAMES) == 5
    ok, f = cap.read()
    assert ok and (f == frames[5]).all()
    assert (cap.read_batch(10) == frames[6:]).all() and cap.read()[0] is False
    assert cap.set(rawvideo.CAP_PROP_POS_FRAMES, 2) and (cap.read_batch(2) == frames[2:4]).all()
    assert not cap.set(rawvideo.CAP_PROP_FPS, 60)
