"""The multi-GPU configurations at their real sizes, rehearsed on ONE MI355X.

  * the full game: 200 000 frames of 1280x720 split into 8 contiguous frame ranges (what 8 ranks own),
    the ranges' (class, max) streams concatenated as the job's all-gather delivers them, segmented once
    - against the same game classified as a single range;
  * throughput mode: 8 clips of 10 000 frames, each on a handle and stream of its own, all in flight
    together - against the clips classified one at a time;
  * the product CLI with --ranks: frame-range sharding of one clip over rank processes (each seeks to
    its range; one all-gather; rank 0 writes) and clips dealt to ranks (no process group; one CSV under
    a file lock) - CSV bytes against the single-process CLI;
  * RCCL itself: a one-rank process group on this GPU runs the job's collective on device tensors;
  * bench.py exactly as the driver starts it for N > 1, with 8 rank processes sharing this GPU.

Full-size cases are checked through properties that do not need the CPU oracle at that size: the class
stream equals the generator's labels, run lengths tile the game, and the sharded CSV equals the
single-range CSV byte for byte.
"""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED = 1234
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cut-detection_amd")


@pytest.fixture(scope="module")
def engine():
    from cutdet import api
    return api.default_engine("cuda:0")


@pytest.fixture(autouse=True)
def _release_hbm():
    """The full-size cases hold most of the GPU's memory; give it back to the driver afterwards (the
    caching allocator would keep it) so that the rank processes other tests start find room."""
    yield
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def _errors(stderr):
    """The lines of a failed child's stderr that name an error, then its tail."""
    hits = [ln for ln in stderr.splitlines() if "Error" in ln or "error" in ln]
    return "\n".join(hits[:20]) + "\n...\n" + stderr[-2000:]


def _cli():
    spec = importlib.util.spec_from_file_location("cd_segment_video", os.path.join(PKG, "segment_video.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _classes(dev, cl):
    return torch.tensor(cl, dtype=torch.uint8, device=dev)


def _need_free(dev, nbytes):
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info(dev)[0]
    assert free >= nbytes, f"needs {nbytes / 1e9:.0f} GB of free HBM, {free / 1e9:.0f} GB available"


# ----------------------------------------------------------------------------- configuration 3 at size
def test_full_game_200k_frames_in_8_contiguous_ranges(engine):
    from cutdet import synth as csynth
    from cutdet.pipeline import ClipClassifier, finish_table, shard_range, TABLE_KEYS
    total, world, h, w, bs = 200_000, 8, 720, 1280, 256
    dev = engine.device
    runs = csynth.run_table(total, SEED)
    cl = csynth.classes_from_runs(runs)
    cls_dev = _classes(dev, cl)
    chunk = 30_000                                    # frames resident at once: 83 GB
    _need_free(dev, chunk * h * w * 3 + (4 << 30))
    frames = torch.empty((chunk, h, w, 3), dtype=torch.uint8, device=dev)

    # (a) as 8 ranks would: rank r classifies shard_range(total, r, 8) from frame 0 of a buffer of its own
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        assert hi - lo == 25_000
        engine.synth_frames(cls_dev[lo:hi], h, w, SEED, first_index=lo, out=frames[: hi - lo])
        rank_clip = ClipClassifier(engine, hi - lo)
        rank_clip.classify_range(frames[: hi - lo], 0, bs)
        parts.append((rank_clip.cls.clone(), rank_clip.maxv.clone()))
    g_cls, g_mx = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])   # what the all-gather delivers
    assert torch.equal(g_cls, cls_dev.to(torch.int64)), "class stream differs from the generator's labels"
    te = engine.rle(g_cls, g_mx)
    sharded = finish_table({k: te[k].cpu().numpy() for k in TABLE_KEYS})
    assert sharded["initial"]["run_lengths"].tolist() == [k for _, k in runs]
    assert sharded["initial"]["frame_types"].tolist() == [c for c, _ in runs]
    assert int(sharded["merged"]["run_lengths"].sum()) == total
    assert sharded["merged"]["start_frames"][0] == 0 and sharded["merged"]["end_frames"][-1] == total - 1
    assert (sharded["merged"]["start_frames"][1:] == sharded["merged"]["end_frames"][:-1] + 1).all()

    # (b) the same game as ONE range (other chunk and batch boundaries): same bits, same CSV
    single = ClipClassifier(engine, total)
    for a in range(0, total, chunk):
        b = min(total, a + chunk)
        engine.synth_frames(cls_dev[a:b], h, w, SEED, first_index=a, out=frames[: b - a])
        single.classify_range(frames[: b - a], a, bs)
    assert torch.equal(single.cls, g_cls) and torch.equal(single.maxv, g_mx)
    res = single.segment()
    assert res["csv"] == sharded["csv"] and len(res["csv"]) > 0
    for k in TABLE_KEYS:
        assert res["merged"][k].tobytes() == sharded["merged"][k].tobytes()
    del frames


# ----------------------------------------------------------------------------- configuration 5 at size
def test_eight_clips_in_flight_match_one_at_a_time(engine):
    """8 clips x 10 000 frames of 1280x720 resident together (221 GB), each classified by a handle of
    its own on a stream of its own, all queued before any is waited for; every clip's CSV must equal
    the one the default handle produces for that clip alone."""
    from cutdet import api, synth as csynth
    from cutdet.engine import Engine
    from cutdet.pipeline import ClipClassifier
    clips, n, h, w, bs = 8, 10_000, 720, 1280, 256
    dev = engine.device
    _need_free(dev, clips * n * h * w * 3 + (8 << 30))
    tables = [csynth.run_table(n, SEED + 17 * c) for c in range(clips)]
    labels = [csynth.classes_from_runs(t) for t in tables]
    frames = [engine.synth_frames(_classes(dev, labels[c]), h, w, SEED + c, first_index=1000 * c) for c in range(clips)]
    engines = [Engine(api.default_blob(), dev) for _ in range(clips)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(clips)]
    work = [ClipClassifier(engines[c], n) for c in range(clips)]
    torch.cuda.synchronize()
    for c in range(clips):                           # everything queued, nothing waited for
        with torch.cuda.stream(streams[c]):
            work[c].classify_range(frames[c], 0, bs)
    torch.cuda.synchronize()
    alone = ClipClassifier(engine, n)
    csvs = set()
    for c in range(clips):
        assert work[c].cls.cpu().tolist() == labels[c]
        alone.classify_range(frames[c], 0, bs, overlap=False)
        assert torch.equal(alone.logits, work[c].logits)
        got, want = work[c].segment(), alone.segment()
        assert got["csv"] == want["csv"]
        assert int(got["merged"]["run_lengths"].sum()) == n
        csvs.add(got["csv"])
    assert len(csvs) == clips                        # eight different clips, eight different results
    for e in engines:
        e.close()
    del frames


# ----------------------------------------------------------------------------- the CLI with --ranks
def _write_gpu_clip(engine, path, n, h, w, seed, first_index=0):
    from cutdet import rawvideo, synth as csynth
    cl = csynth.classes_from_runs(csynth.run_table(n, seed))
    frames = engine.synth_frames(_classes(engine.device, cl), h, w, SEED, first_index=first_index)
    rawvideo.write_raw_video(path, frames.cpu().numpy(), fps=30)
    return cl


def test_cli_ranks_shard_one_clip_by_frame_range(engine, tmp_path):
    """segment_video.py --ranks N on one GPU (gloo): every rank seeks to its contiguous range, one
    all-gather, rank 0 writes.  CSV bytes must equal the single-process CLI's, also with a frame limit
    and a batch size that does not divide the ranges."""
    mod = _cli()
    clip = str(tmp_path / "game.cdraw")
    _write_gpu_clip(engine, clip, 3001, 180, 320, SEED)
    one = mod.main(mod.sv_parser.parse_args([clip, "--output_path", str(tmp_path / "one.csv")]))
    want = open(one, "rb").read()
    assert want.count(b"\r\n") >= 3
    two = mod.main(mod.sv_parser.parse_args([clip, "--output_path", str(tmp_path / "two.csv"), "--ranks", "2",
                                             "--rank-devices", "0,0"]))
    assert open(two, "rb").read() == want
    lim1 = mod.main(mod.sv_parser.parse_args([clip, "--output_path", str(tmp_path / "lim1.csv"), "--frame-limit", "1500",
                                              "--batch-size", "100", "--columns", "start,type,end_frame"]))
    lim3 = mod.main(mod.sv_parser.parse_args([clip, "--output_path", str(tmp_path / "lim3.csv"), "--frame-limit", "1500",
                                              "--batch-size", "100", "--columns", "start,type,end_frame",
                                              "--ranks", "3", "--rank-devices", "0,0,0"]))
    got = open(lim3, "rb").read()
    assert got == open(lim1, "rb").read()
    assert got.strip().split(b"\r\n")[-1].split(b",")[2] == b"1599"      # the batch that crosses the limit is scored
    with pytest.raises(ValueError):
        mod.main(mod.sv_parser.parse_args([clip, "--ranks", "2", "--rank-devices", "0"]))


def test_cli_ranks_deal_clips_one_per_gpu(engine, tmp_path):
    """8 clips, 8 rank processes (sharing this GPU), no process group: every rank appends its clip to
    the one CSV under a file lock.  The file must hold exactly the single-process rows of every clip,
    each clip's rows in one piece."""
    mod = _cli()
    paths = []
    for c in range(8):
        p = str(tmp_path / f"clip{c}.cdraw")
        _write_gpu_clip(engine, p, 700 + 37 * c, 180, 320, SEED + c, first_index=100 * c)
        paths.append(p)
    seq = mod.main(mod.sv_parser.parse_args([paths[0], "--more-inputs", *paths[1:], "--output_path",
                                             str(tmp_path / "seq.csv")]))
    par = mod.main(mod.sv_parser.parse_args([paths[0], "--more-inputs", *paths[1:], "--output_path",
                                             str(tmp_path / "par.csv"), "--ranks", "8", "--rank-devices",
                                             ",".join(["0"] * 8)]))

    def blocks(path):
        out, order = {}, []
        for line in open(path, "rb").read().split(b"\r\n"):
            if line:
                name = line.split(b",")[0]
                if not order or order[-1] != name:
                    order.append(name)
                out.setdefault(name, []).append(line)
        return out, order

    want, order_seq = blocks(seq)
    got, order_par = blocks(par)
    assert order_seq == [os.path.basename(p).encode() for p in paths]
    assert sorted(order_par) == sorted(order_seq), "a clip's rows were split or a clip is missing"
    assert got == want


# ----------------------------------------------------------------------------- RCCL on this GPU
RCCL_ONE_RANK = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from cutdet.pipeline import gather_streams
cls = torch.arange(1001, device=dev) % 3
mx = torch.rand(1001, device=dev)
calls = []
real = dist.all_gather_into_tensor
dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
g_cls, g_mx = gather_streams(cls, mx, dist.group.WORLD, 1001, force_collective=True)
torch.cuda.synchronize()
assert len(calls) == 1 and g_cls.is_cuda and torch.equal(g_cls, cls) and torch.equal(g_mx, mx)
print("RCCL_OK", dist.get_backend(), torch.cuda.nccl.version())
dist.destroy_process_group()
"""


def test_rccl_runs_the_stream_gather_on_device_tensors():
    """The job's collective through RCCL itself: a one-rank "nccl" process group on this GPU gathers
    device tensors (more ranks need more GPUs: two ranks on one GPU is refused by RCCL)."""
    from cutdet.launch import free_port
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", RCCL_ONE_RANK, PKG, str(free_port())], env=env, capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0 and "RCCL_OK nccl" in r.stdout, (r.stdout + r.stderr)[-3000:]


# ----------------------------------------------------------------------------- bench.py as the driver starts it
def test_bench_driver_command_with_8_ranks_reports_all_three_shapes():
    """`python bench.py --gpus 8` (no other shape flags): the weak headline plus the strong 200 000-frame
    game and the independent clips, each with value / ms_per_step / n_ranks_seen.  Eight rank processes
    share this GPU (gloo stands in for RCCL); frames beyond each rank's share of HBM are regenerated
    outside the timed intervals."""
    e = dict(os.environ, CUTDET_BENCH_BACKEND="gloo", CUTDET_BENCH_ONE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
                        "--leg-steps", "1", "--leg-warmup", "1"], env=e, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, _errors(r.stderr)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and lines[0] == r.stdout.strip()
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["n_ranks_seen"] == 8
    assert out["config"]["frames_per_step"] == 80_000 and out["value"] > 0
    s = out["strong_200k"]
    assert s["n_ranks_seen"] == 8 and s["frames_per_step"] == 200_000 and s["frames_per_rank"] == 25_000
    assert s["value"] > 0 and s["ms_per_step"] > 0 and s["segments_out"] > 100 and "one all-gather" in s["stitch"]
    c = out["independent_clips"]
    assert c["n_ranks_seen"] == 8 and c["clips"] == 8 and c["process_group"] == "none"
    assert c["value"] > 0 and c["ms_per_step"] > 0 and c["frames_per_step"] == 80_000
    assert out["roofline"]["launches_timed"] > 0 and 0 < out["roofline"]["frac"] < 1


def test_bench_one_gpu_line_has_config4_and_the_separate_roofline_pass():
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=e, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, _errors(r.stderr)
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 1 and out["dtype"] == "f32" and out["config"]["frames_per_step"] == 10_000
    assert out["roofline"]["bound"] == "mfma" and out["roofline"]["launches_timed"] == 80
    assert "separate single-stream pass" in out["roofline"]["measured_in"]
    c4 = out["config4"]
    assert "1920x1080" in c4["workload"] and c4["roofline"]["bound"] == "hbm" and c4["roofline"]["peak"] == 8000.0
    assert c4["value"] > 0 and 0 < c4["roofline"]["frac"] < 1
    assert abs(c4["roofline"]["achieved"] - 1_658_892 * c4["value"] / 1e9) < 1.0
    assert set(c4["stages_single_stream_pass"]) >= {"conv1", "conv2", "conv3_head"}


def test_bench_under_torch_distributed_run_with_2_ranks():
    """The driver's other launch form: `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`.
    No launcher directory exists then: rank 0 makes the one the file barrier of the independent-clips leg
    uses and passes it round through the process group before that group is destroyed."""
    from cutdet.launch import free_port
    e = dict(os.environ, CUTDET_BENCH_BACKEND="gloo", CUTDET_BENCH_ONE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CUTDET_RUN_DIR"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--leg-steps", "1", "--leg-warmup", "1"],
                       env=e, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, _errors(r.stderr)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["n_ranks_seen"] == 2 and out["config"]["frames_per_step"] == 20_000
    assert out["strong_200k"]["n_ranks_seen"] == 2 and out["strong_200k"]["frames_per_rank"] == 100_000
    assert "intervals" in out["strong_200k"]["timed_region"]          # 100 000 frames per rank do not fit: chunked
    assert out["independent_clips"]["n_ranks_seen"] == 2 and out["independent_clips"]["process_group"] == "none"
