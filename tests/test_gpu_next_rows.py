"""GPU tests of the rows next to the hot path: model files and training-side validation, the CSV
wire format through the CLI, and scale cases of the run-length pass.  Needs an MI355X."""
import csv
import importlib.util
import io
import json
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED = 1234
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def engine():
    from cutdet import api
    return api.default_engine("cuda:0")


def _cli():
    spec = importlib.util.spec_from_file_location("cd_segment_video",
                                                  os.path.join(ROOT, "cut-detection_amd", "segment_video.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write_clip(tmp_path, name, runs, h, w, first_index=0):
    from cutdet import rawvideo, synth as csynth
    from oracle import synth
    cl = csynth.classes_from_runs(runs)
    frames = synth.clip(SEED, cl, h, w, first_index=first_index)
    p = str(tmp_path / name)
    rawvideo.write_raw_video(p, frames, fps=30)
    return p, frames, cl


# ----------------------------------------------------------------------------- model files
@pytest.mark.parametrize("case", ["720p", "180p"])
def test_load_and_glue_nets_on_the_reference_files_gives_golden_logits(golden_dir, case):
    """frameID/net.py:193-217 on the reference's own three files, then the GPU forward, against
    logits the reference itself produced (net_golden.npz)."""
    from frameID.net import load_and_glue_nets
    from oracle import synth
    d = os.path.join(golden_dir, "ref_prod_net")
    net, params = load_and_glue_nets(os.path.join(d, "init_model_model_params.json"),
                                     os.path.join(d, "init_model_classifier_conv.pt"),
                                     os.path.join(d, "init_model_classifier_linear.pt"))
    net.eval()
    net.to("cuda:0")
    g = np.load(os.path.join(golden_dir, "net_golden.npz"))
    with open(os.path.join(golden_dir, "net_golden_meta.json")) as f:
        m = json.load(f)[case]
    frames = torch.from_numpy(synth.clip(SEED, m["classes"], m["height"], m["width"])).to("cuda:0")
    logits, _, am = net.classify_frames(frames)
    assert np.abs(logits.cpu().numpy() - g[f"logits_{case}"]).max() <= 1e-3
    assert am.cpu().tolist() == m["classes"]
    # the float entry the reference's driver uses: net(batch).  Its block 1 runs on the fp32 matrix instructions
    # (float frames are not bytes); the u8 entry agrees with it bit for bit under the same setting and to a few
    # ulp with its default block 1 (three exact weight terms on the bf16 pipe)
    y = net(net.engine.preprocess(frames)).cpu().numpy()
    assert np.abs(y - logits.cpu().numpy()).max() <= 2e-5
    net.engine.set_option("f32_block1_fp32_mfma", 1)
    assert y.tobytes() == net.classify_frames(frames)[0].cpu().numpy().tobytes()
    net.engine.set_option("f32_block1_fp32_mfma", 0)


def test_evaluate_matches_the_validation_loop(engine):
    """ProdNet.evaluate against the oracle's restatement of supervised_training.py:177-215 on
    synthetic frames with a few deliberately wrong labels: counts exact, mean loss to 1e-5."""
    from frameID.net import load_default_net
    from oracle import evaluate, net as onet, preprocess, synth
    net, _ = load_default_net()
    net.eval().to(engine.device)
    cl = [0] * 20 + [1] * 17 + [2] * 11 + [1] * 9 + [0] * 13           # 70 frames
    labels = np.array(cl)
    labels[[3, 25, 40, 69]] = [1, 2, 0, 2]                              # mislabelled frames
    frames = synth.clip(SEED, cl, 180, 320)
    w = onet.load_blob()
    bounds = [(0, 32), (32, 64), (64, 70)]
    want = evaluate.validate([onet.forward(w, preprocess.preprocess_batch(frames[a:b])).numpy() for a, b in bounds],
                             [labels[a:b] for a, b in bounds])
    dev_frames = torch.from_numpy(frames).to(engine.device)
    # u8 frames as (x, y) pairs with [B,1] labels (the reference's loader shape, squeezed by the script)
    got = net.evaluate((dev_frames[a:b], torch.from_numpy(labels[a:b]).reshape(-1, 1)) for a, b in bounds)
    assert got["correct"] == want["correct"] and got["total"] == want["total"] and got["frames"] == 70
    assert got["correct"] == [31, 25, 10] and got["total"] == [32, 26, 12]
    assert abs(got["loss"] - want["loss"]) <= 1e-5
    assert got["accuracy"] == pytest.approx(want["accuracy"])
    # float frames as the reference loader's dicts give the same counts
    x = engine.preprocess(dev_frames)
    got2 = net.evaluate({"x": x[a:b], "y": torch.from_numpy(labels[a:b])} for a, b in bounds)
    assert got2["correct"] == got["correct"] and abs(got2["loss"] - got["loss"]) <= 1e-6   # block 1: other order of additions
    # labels outside the class range are skipped
    bad = torch.tensor([0, 7, -1, 2])
    got3 = net.evaluate([(dev_frames[[0, 1, 2, 40]], bad)])
    assert got3["total"] == [1, 0, 1] and got3["frames"] == 2


# ----------------------------------------------------------------------------- wire format through the CLI
def test_cli_columns_and_multi_clip_stream(engine, tmp_path):
    from oracle import net as onet, preprocess, segmentation
    mod = _cli()
    w = onet.load_blob()
    clips = [("a.cdraw", [(0, 130), (2, 15), (1, 140)], 0), ("b.cdraw", [(1, 150), (2, 5), (1, 60), (0, 120)], 500)]
    paths, want = [], {}
    for name, runs, first in clips:
        p, frames, cl = _write_clip(tmp_path, name, runs, 180, 320, first)
        paths.append(p)
        scores = np.concatenate([onet.forward(w, preprocess.preprocess_batch(frames[a:a + 100])).numpy()
                                 for a in range(0, len(cl), 100)])
        want[name] = segmentation.segment_clip(scores)
    # one clip, extra columns: the first two fields of every row are the reference's CSV
    out = mod.main(mod.sv_parser.parse_args([paths[0], "--output_path", str(tmp_path / "wide.csv"), "--columns",
                                             "start,type,end_frame,run_length,score_mean", "--batch-size", "64"]))
    rows = list(csv.reader(io.StringIO(open(out, newline="").read(), newline="")))
    merged = want["a.cdraw"][2]
    assert ("\r\n".join(",".join(r[:2]) for r in rows) + "\r\n").encode() == want["a.cdraw"][3]
    assert [int(r[2]) for r in rows] == merged["end_frames"] and [int(r[3]) for r in rows] == merged["run_lengths"]
    np.testing.assert_allclose([float(r[4]) for r in rows], np.array(merged["score_means"], dtype=np.float64), rtol=2e-6)
    # default columns: byte-identical to the reference's writer
    out = mod.main(mod.sv_parser.parse_args([paths[0], "--output_path", str(tmp_path / "plain.csv")]))
    assert open(out, "rb").read() == want["a.cdraw"][3]
    # two clips: one streamed file with a clip column
    out = mod.main(mod.sv_parser.parse_args([paths[0], "--more-inputs", paths[1], "--output_path",
                                             str(tmp_path / "both.csv"), "--batch-size", "64"]))
    data = open(out, "rb").read()
    expect = b"".join(b"".join(name.encode() + b"," + line + b"\r\n" for line in want[name][3].split(b"\r\n") if line)
                      for name, _, _ in clips)
    assert data == expect


def test_cli_keep_lone_orphan(engine, tmp_path):
    mod = _cli()
    p, _, _ = _write_clip(tmp_path, "short.cdraw", [(0, 75)], 180, 320)       # one shot, shorter than 100 frames
    with pytest.raises(IndexError):
        mod.main(mod.sv_parser.parse_args([p]))
    out = mod.main(mod.sv_parser.parse_args([p, "--keep-lone-orphan", "--columns", "start,type,end_frame"]))
    rows = open(out, "rb").read().split(b"\r\n")
    assert len(rows) == 2 and rows[0].split(b",")[0] == b"0" and rows[0].split(b",")[2] == b"74"


# ----------------------------------------------------------------------------- run-length scale cases
def test_rle_one_16m_frame_segment(engine):
    """A single segment of 2^24 frames: exact table, mean = the fp64 sum rounded once, and - the sum
    coming from per-block partial sums instead of one wavefront walking the segment - within 5 ms."""
    n = 1 << 24
    dev = engine.device
    cls = torch.ones((n,), dtype=torch.int64, device=dev)
    mx = torch.rand((n,), dtype=torch.float32, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 8 + 1
    engine.rle(cls, mx, frame_offset=5, cap=64)             # warm the kernels and the allocator
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    te = engine.rle(cls, mx, frame_offset=5, cap=64)        # includes the read-back of the segment count
    dt = time.perf_counter() - t0
    assert te["end_frames"].cpu().tolist() == [n - 1 + 5] and te["start_frames"].cpu().tolist() == [5]
    assert te["run_lengths"].cpu().tolist() == [n] and te["frame_types"].cpu().tolist() == [1]
    want = mx.double().sum().item()
    assert abs(te["score_sums"].item() - want) <= 1e-9 * want
    assert te["score_means"].item() == np.float32(want / n)
    assert dt < 5e-3, f"{dt * 1e3:.2f} ms"
    print(f"rle of one 2^24-frame segment: {dt*1e3:.1f} ms")
    # two long segments with a single-frame one between them
    cls[n // 2] = 2
    te = engine.rle(cls, mx)
    assert te["run_lengths"].cpu().tolist() == [n // 2, 1, n - n // 2 - 1]
    assert te["score_means"][1].item() == mx[n // 2].item()


# ----------------------------------------------------------------------------- NV12 ingest
def _bgr_to_nv12(frames):
    """Test helper: BT.601 limited-range NV12 planes [n, H*3/2, W] from BGR frames (float math,
    2x2 chroma average); only used to make plausible NV12 input."""
    f = frames.astype(np.float64)
    b, g, r = f[..., 0], f[..., 1], f[..., 2]
    y = 16 + 0.257 * r + 0.504 * g + 0.098 * b
    u = 128 - 0.148 * r - 0.291 * g + 0.439 * b
    v = 128 + 0.439 * r - 0.368 * g - 0.071 * b
    n, h, w = y.shape
    pool = lambda a: a.reshape(n, h // 2, 2, w // 2, 2).mean(axis=(2, 4))  # noqa: E731
    uv = np.empty((n, h // 2, w), np.float64)
    uv[:, :, 0::2], uv[:, :, 1::2] = pool(u), pool(v)
    return np.clip(np.rint(np.concatenate([y, uv], axis=1)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w,order", [(720, 1280, 0), (36, 62, 1), (8, 6, 0)])
def test_nv12_conversion_is_bit_exact(engine, h, w, order):
    from oracle import preprocess
    rs = np.random.RandomState(h + w)
    planes = rs.randint(0, 256, size=(3, h * 3 // 2, w)).astype(np.uint8)      # every byte value, every combination
    dev = torch.from_numpy(planes).to(engine.device)
    got = engine.nv12_to_bgr(dev[:, :h], dev[:, h:], chan_order=order).cpu().numpy()
    want = np.stack([preprocess.nv12_to_bgr(p[:h], p[h:], chan_order=order) for p in planes])
    assert got.shape == (3, h, w, 3) and (got == want).all()
    # selected rows, from planes with padded row strides (views into wider buffers)
    yr = np.array(sorted(rs.choice(h, size=max(2, h // 5), replace=False)), dtype=np.int32)
    ur = (yr // 2).astype(np.int32)
    wide = torch.zeros((3, h * 3 // 2, w + 10), dtype=torch.uint8, device=engine.device)
    wide[:, :, 1:w + 1] = dev
    view = wide[:, :, 1:w + 1]                                                 # rows start at an odd address
    got = engine.nv12_to_bgr(view[:, :h], view[:, h:], torch.from_numpy(yr).to(engine.device),
                             torch.from_numpy(ur).to(engine.device), chan_order=order).cpu().numpy()
    want = np.stack([preprocess.nv12_to_bgr(p[:h], p[h:], yr, ur, chan_order=order) for p in planes])
    assert (got == want).all()


def test_nv12_container_end_to_end(engine, tmp_path):
    """A 720p NV12 clip: sampled-row plane upload + GPU conversion + classify gives the logits of
    converting whole frames with the oracle and classifying those; the CLI segments it."""
    from cutdet import rawvideo, synth as csynth
    from frameID.data import VideoDataset
    from frameID.net import load_default_net
    from oracle import net as onet, preprocess, segmentation, synth
    runs = [(0, 110), (2, 12), (1, 120)]
    cl = csynth.classes_from_runs(runs)
    planes = _bgr_to_nv12(synth.clip(SEED, cl, 720, 1280))
    path = str(tmp_path / "clip_nv12.cdraw")
    rawvideo.write_raw_video(path, planes, chan_order=rawvideo.NV12)
    bgr = np.stack([preprocess.nv12_to_bgr(p[:720], p[720:]) for p in planes])
    w = onet.load_blob()
    want = np.concatenate([onet.forward(w, preprocess.preprocess_batch(bgr[a:a + 22])).numpy()
                           for a in range(0, len(cl), 22)])
    net, _ = load_default_net()
    net.eval().to(engine.device)
    for sampled in (True, False):
        ds = VideoDataset(path, resize=256, device=engine.device)
        assert ds.nv12 and ds.video_info == {"fps": 30, "length": len(cl), "width": 1280, "height": 720}
        got = []
        for b in ds.frame_batches(64, sampled=sampled):
            assert tuple(b.shape[1:]) == ((144 if sampled else 720), 1280, 3)
            got.append(net.classify_frames(b, chan_order=ds.chan_order, net_h=ds.batch_net_height)[0])
        got = torch.cat(got).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-3
    assert want.argmax(1).tolist() == cl                      # the round trip through NV12 keeps the classes
    mod = _cli()
    out = mod.main(mod.sv_parser.parse_args([path, "--batch-size", "64"]))
    assert open(out, "rb").read() == segmentation.segment_clip(want)[3]
    # the reference-style iterator over an NV12 container yields the float frames of the converted clip
    it = iter(VideoDataset(path, resize=256, device=engine.device))
    first = next(it).cpu().numpy()
    assert first.tobytes() == preprocess.preprocess(bgr[0]).tobytes()


# ----------------------------------------------------------------------------- multi-rank shapes on one GPU
def _bench_json(args, **env):
    import subprocess
    import sys
    e = dict(os.environ, CUTDET_BENCH_BACKEND="gloo", CUTDET_BENCH_ONE_GPU="1", **env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                       timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and lines[0] == r.stdout.strip(), "stdout must hold the one JSON line and nothing else"
    return json.loads(lines[0])


def test_bench_full_game_shape_in_contiguous_ranges():
    """The frame-range sharded shape (a game split into contiguous ranges, one all-gather of the
    (class, max) streams, rank 0 segments the whole game) with two rank processes sharing this GPU
    over gloo: the job must segment the game exactly as one process does."""
    one = _bench_json(["--total-frames", "6000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    two = _bench_json(["--gpus", "2", "--total-frames", "6000", "--steps", "1", "--warmup", "1"])
    assert two["n_gpus"] == 2 and two["config"]["n_ranks_seen"] == 2 and two["scaling"] == "strong"
    assert two["config"]["frames_per_step"] == 6000 and two["config"]["frames_per_gpu_timed"] == 3000
    assert "one all-gather" in two["config"]["stitch"]
    assert two["config"]["segments_out"] == one["config"]["segments_out"] > 1
    # a range longer than the resident budget is classified in chunks, regeneration untimed
    chunked = _bench_json(["--total-frames", "6000", "--resident-frames", "2048", "--steps", "1", "--warmup", "1",
                           "--no-cpu-baseline"])
    assert chunked["config"]["segments_out"] == one["config"]["segments_out"]
    assert "intervals" in chunked["config"]["timed_region"] and one["config"]["timed_region"] == "one interval"


def test_bench_independent_clips_shape():
    """One clip per rank, no collective: two rank processes on this GPU."""
    out = _bench_json(["--gpus", "2", "--independent-clips", "--clip-frames", "2500", "--steps", "1", "--warmup", "1"])
    assert out["config"]["stitch"] == "none" and out["config"]["n_ranks_seen"] == 2
    assert out["config"]["frames_per_step"] == 5000 and out["scaling"] == "weak"


# ----------------------------------------------------------------------------- stand-alone resize, overlapped pipeline
@pytest.mark.parametrize("h,w", [(1080, 1920), (180, 320), (480, 640), (512, 512), (100, 300), (720, 1280), (90, 161)])
def test_resize_u8_is_bit_exact(engine, h, w):
    from oracle import preprocess, synth
    frames = synth.clip(SEED, [0, 1, 2], h, w)
    nh, nw = preprocess.output_size(h, w)
    want = np.stack([preprocess.resize_linear_u8(f, nh, nw) for f in frames])
    got = engine.resize_u8(torch.from_numpy(frames).to(engine.device)).cpu().numpy()
    assert got.shape == want.shape and (got == want).all()


@pytest.mark.parametrize("h,w,resize", [(1080, 1920, 320), (90, 161, 320), (360, 640, 320), (200, 333, 100), (64, 96, 48),
                                        (50, 77, 30), (9, 2, 4)])
def test_resize_kernels_agree_on_any_row_layout(h, w, resize):
    """The resize has two kernels: taps straight from global memory (default; no alignment needs) and source
    rows staged in LDS (CD_OPT_RESIZE_ROWS).  Both give the oracle's bytes - bilinear, point-sampled and 2x2
    box geometries, right-edge taps, a last row group that is not full, output widths that are not a multiple
    of 4 and rows shorter than 8 bytes (where the default kernel hands over) - also on views whose rows start
    at odd addresses and are padded."""
    from cutdet import api
    from cutdet.engine import Engine
    from oracle import preprocess, synth
    base = api.default_engine("cuda:0", "f32")
    rows = Engine(api.default_blob(), base.device, "f32")
    rows.set_option("resize_rows", 1)
    frames = synth.clip(SEED + 5, [0, 1, 2, 1], h, w)
    nh, nw = preprocess.output_size(h, w, resize)
    want = np.stack([preprocess.resize_linear_u8(f, nh, nw) for f in frames])
    dev = torch.from_numpy(frames).to(base.device)
    wide = torch.zeros((4, h + 1, w + 7, 3), dtype=torch.uint8, device=base.device)
    view = wide[:, 1:, 3:w + 3]
    view.copy_(dev)
    for eng in (base, rows):
        for src in (dev, view):
            got = eng.resize_u8(src, resize).cpu().numpy()
            assert got.shape == want.shape and (got == want).all()
    rows.close()


@pytest.mark.parametrize("h,w,dtype", [(1080, 1920, "f32"), (180, 320, "f32"), (1080, 1920, "bf16_split"), (180, 320, "bf16")])
def test_overlapped_resize_pipeline_matches_the_fused_call(h, w, dtype):
    """classify_range with the resize on a second stream (three buffer reuses, a short last batch)
    gives the bits of one fused call per batch, in every arithmetic."""
    from cutdet import api, synth as csynth
    from cutdet.pipeline import ClipClassifier
    eng = api.default_engine("cuda:0", dtype)
    n = 150
    cl = csynth.classes_from_runs(csynth.run_table(n, SEED))
    frames = eng.synth_frames(torch.tensor(cl, dtype=torch.uint8, device=eng.device), h, w, SEED)
    fused, piped = ClipClassifier(eng, n), ClipClassifier(eng, n)
    fused.classify_range(frames, 0, 32, overlap=False)
    for _ in range(2):                                   # twice: the buffers and events are reused across calls
        piped.logits.fill_(float("nan"))
        piped.classify_range(frames, 0, 32)
        torch.cuda.synchronize()
        assert torch.equal(piped.logits, fused.logits)
        assert torch.equal(piped.cls, fused.cls) and torch.equal(piped.maxv, fused.maxv)
    assert fused.cls.cpu().tolist() == cl


@pytest.mark.parametrize("dtype", ["f32", "bf16_split"])
def test_range_entry_matches_single_batch_calls(dtype):
    """cd_forward_u8_range (batches alternating between two workspaces on two streams inside the
    library) gives the bits of cd_forward_u8 batch by batch, repeatedly, with a short last batch, and
    its results are ordered on the caller's stream when it returns."""
    from cutdet import api, synth as csynth
    from cutdet.pipeline import ClipClassifier
    eng = api.default_engine("cuda:0", dtype)
    n = 700
    cl = csynth.classes_from_runs(csynth.run_table(n, SEED))
    frames = eng.synth_frames(torch.tensor(cl, dtype=torch.uint8, device=eng.device), 720, 1280, SEED)
    one = ClipClassifier(eng, n)
    one.classify_range(frames, 0, 128, overlap=False)
    two = ClipClassifier(eng, n)
    for _ in range(3):
        two.logits.fill_(float("nan"))
        two.classify_range(frames, 0, 128)          # 6 batches, the last one short
        assert torch.equal(two.logits, one.logits) and torch.equal(two.cls, one.cls)   # same stream: ordered after the join
        assert torch.equal(two.maxv, one.maxv)
    assert two.segment()["csv"] == one.segment()["csv"]
    # a range of one batch, and a range at an offset into the clip buffers
    two.logits.fill_(float("nan"))
    two.classify_range(frames[:100], 0, 128)
    two.classify_range(frames[100:], 100, 256)
    assert torch.equal(two.logits, one.logits)
    # on a non-default stream
    st = torch.cuda.Stream(device=eng.device)
    st.wait_stream(torch.cuda.current_stream(eng.device))
    with torch.cuda.stream(st):
        two.logits.fill_(float("nan"))
        two.classify_range(frames, 0, 64)
        same = torch.equal(two.logits, one.logits)
    st.synchronize()
    assert same


@pytest.mark.parametrize("h,w,bs,n", [(720, 1280, 96, 500), (1080, 1920, 64, 200), (180, 320, 50, 333), (480, 640, 40, 90)])
def test_fused_blocks_1_and_2_give_the_bits_of_the_separate_kernels(h, w, bs, n):
    """Split bf16 with CD_OPT_FUSE_BLOCKS12: the resize pass writes operand cells and blocks 1 + 2 run as
    one kernel (producer and consumer waves, block-1 activations in LDS).  Logits, maxima and classes
    must equal the per-batch calls' bit for bit: point-sampling and bilinear geometries, a 4:3 frame
    (192 network rows: other tile counts and frame margins), short last batches, repeated calls."""
    from cutdet import api, synth as csynth
    from cutdet.engine import Engine
    from cutdet.pipeline import ClipClassifier
    base = api.default_engine("cuda:0", "bf16_split")
    eng = Engine(api.default_blob(), base.device, "bf16_split")
    eng.set_option("fuse_blocks12", 1)
    cl = csynth.classes_from_runs(csynth.run_table(n, SEED))
    frames = base.synth_frames(torch.tensor(cl, dtype=torch.uint8, device=base.device), h, w, SEED)
    want = ClipClassifier(base, n)
    want.classify_range(frames, 0, bs, overlap=False)
    got = ClipClassifier(eng, n)
    for _ in range(2):
        got.logits.fill_(float("nan"))
        got.classify_range(frames, 0, bs)
        assert torch.equal(got.logits, want.logits) and torch.equal(got.cls, want.cls) and torch.equal(got.maxv, want.maxv)
    assert got.cls.cpu().tolist() == cl
    # the other schedules of the range entry on the same handle
    for opt in ("range_one_stream", "range_by_stage"):
        eng.set_option("fuse_blocks12", 0)
        eng.set_option(opt, 1)
        got.logits.fill_(float("nan"))
        got.classify_range(frames, 0, bs)
        assert torch.equal(got.logits, want.logits), opt
        eng.set_option(opt, 0)
    from cutdet import _lib
    assert _lib.lib.cd_set_option(eng._h, 9, 1) == _lib.CD_EINVAL and _lib.lib.cd_set_option(eng._h, 1, 2) == _lib.CD_EINVAL
    eng.close()


@pytest.mark.parametrize("h,w,n", [(720, 1280, 300), (180, 320, 77), (480, 640, 130), (90, 160, 9)])
def test_split_block2_two_waves_per_simd_gives_the_bits_of_the_one_wave_kernel(h, w, n):
    """CD_OPT_SPLIT_TWO_WAVES: split bf16 block 2 with two waves per SIMD (one patch buffer, copied over
    during a unit's last step, rows of a step one after the other).  Same logits bit for bit as the default
    one-wave kernel, for batches that fill the grid many times over and for ones smaller than it, repeated calls."""
    from cutdet import api, synth as csynth
    from cutdet.engine import Engine
    from cutdet.pipeline import ClipClassifier
    one = api.default_engine("cuda:0", "bf16_split")
    base = Engine(api.default_blob(), one.device, "bf16_split")
    base.set_option("split_two_waves", 1)
    cl = csynth.classes_from_runs(csynth.run_table(n, SEED + 3))
    frames = base.synth_frames(torch.tensor(cl, dtype=torch.uint8, device=base.device), h, w, SEED + 3)
    want = ClipClassifier(one, n)
    want.classify_range(frames, 0, 128, overlap=False)
    got = ClipClassifier(base, n)
    for bs in (128, 7, n):
        got.logits.fill_(float("nan"))
        got.classify_range(frames, 0, bs, overlap=False)
        assert torch.equal(got.logits, want.logits) and torch.equal(got.cls, want.cls) and torch.equal(got.maxv, want.maxv), bs
    got.logits.fill_(float("nan"))
    got.classify_range(frames, 0, 64)                 # the two-stream range entry
    assert torch.equal(got.logits, want.logits)
    assert got.cls.cpu().tolist() == cl
    base.close()


def test_range_entry_rejects_bad_arguments(engine):
    from cutdet._lib import CutdetError
    dev = engine.device
    frames = torch.zeros((4, 180, 320, 3), dtype=torch.uint8, device=dev)
    outs = (torch.zeros((4, 3), device=dev), torch.zeros((4,), device=dev), torch.zeros((4,), dtype=torch.int64, device=dev))
    with pytest.raises(CutdetError):
        engine.forward_u8_range(frames, 0, outs)                   # batch of 0
    with pytest.raises(ValueError):
        engine.forward_u8_range(frames, 2, (outs[0][:3], outs[1], outs[2]))
    engine.forward_u8_range(frames[:0], 2, (outs[0][:0], outs[1][:0], outs[2][:0]))      # empty range: no-op


# ----------------------------------------------------------------------------- argument checks of the newer entry points
def test_new_entry_points_reject_bad_arguments(engine):
    from cutdet._lib import CutdetError
    dev = engine.device
    y = torch.zeros((1, 8, 7), dtype=torch.uint8, device=dev)               # odd width
    with pytest.raises(CutdetError):
        engine.nv12_to_bgr(y, torch.zeros((1, 4, 7), dtype=torch.uint8, device=dev))
    with pytest.raises(ValueError):
        engine.nv12_to_bgr(torch.zeros((1, 8, 8), dtype=torch.uint8, device=dev),
                           torch.zeros((1, 3, 8), dtype=torch.uint8, device=dev))    # chroma plane too short
    with pytest.raises(RuntimeError):
        engine.resize_u8(torch.zeros((1, 8, 8, 3), dtype=torch.uint8))     # CPU tensor: no fallback
    with pytest.raises(ValueError):
        engine.resize_u8(torch.zeros((2, 90, 160, 3), dtype=torch.uint8, device=dev),
                         out=torch.zeros((2, 100, 256, 3), dtype=torch.uint8, device=dev))   # wrong output shape
    logits = torch.zeros((4, 9), device=dev)
    nine = torch.zeros((9,), dtype=torch.int64, device=dev)
    with pytest.raises(CutdetError):                                       # more than 8 classes
        engine.eval_accumulate(logits, torch.zeros((4,), dtype=torch.int64, device=dev), nine, nine.clone(),
                               torch.zeros((1,), dtype=torch.float64, device=dev))
    with pytest.raises(ValueError):
        engine.eval_accumulate(torch.zeros((4, 3), device=dev), torch.zeros((5,), dtype=torch.int64, device=dev),
                               nine[:3], nine[:3].clone(), torch.zeros((1,), dtype=torch.float64, device=dev))
    # an empty batch is a no-op for the wrappers, an error at the ABI
    assert engine.resize_u8(torch.zeros((0, 90, 160, 3), dtype=torch.uint8, device=dev)).shape == (0, 144, 256, 3)


# ----------------------------------------------------------------------------- TorchScript consumer
def test_forward_traces_as_one_operator_and_survives_save_and_load(engine, tmp_path):
    """The reference's exporter (training_scripts/make_torchscript_model.py:25-34): trace on a random
    [1,3,144,256] input, compare traced with eager, save.  Here the forward is one registered operator;
    the traced module must give eager's bits, on other batch sizes too, also after a save / load
    round trip in a fresh process that has only imported cutdet.torchscript."""
    import subprocess
    import sys
    from cutdet import torchscript as ts
    traced, eager = ts.trace("f32", engine.device)
    assert "cutdet::prod_net_forward" in str(traced.graph)
    x = torch.rand((5, 3, 144, 256), device=engine.device)
    want = engine.forward(x)
    assert torch.equal(eager(x), want) and torch.equal(traced(x), want)
    path = str(tmp_path / "saved_model_trace.pt")
    traced.save(path)
    torch.save(x.cpu(), str(tmp_path / "x.pt"))
    torch.save(want.cpu(), str(tmp_path / "y.pt"))
    code = ("import sys, torch; sys.path.insert(0, sys.argv[1]); import cutdet.torchscript; "
            "m = torch.jit.load(sys.argv[2]); x = torch.load(sys.argv[3]).cuda(); "
            "y = m(x).cpu(); assert torch.equal(y, torch.load(sys.argv[4])); print('LOADED_OK')")
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "cut-detection_amd"), path,
                        str(tmp_path / "x.pt"), str(tmp_path / "y.pt")], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "LOADED_OK" in r.stdout, (r.stdout + r.stderr)[-2000:]
