"""HIP path vs the CPU oracle and the reference-generated fixtures.  Needs an MI355X.

Bars: bytes, indices, class ids and segment tables bit-exact; fp32 logits within 1e-3 of the
reference's CPU logits (the tolerance the task states); per-segment score means within 1e-6
relative (the reference sums them in fp32 in an unspecified order, the GPU in fp64).
"""
import base64
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED = 1234
LOGIT_TOL = 1e-3


@pytest.fixture(scope="module")
def engine():
    from cutdet import api
    return api.default_engine("cuda:0")


@pytest.fixture(scope="module")
def fp32_mfma_engine():
    """fp32 with block 1 on the fp32 matrix instructions as well (CD_OPT_F32_BLOCK1_FP32_MFMA): the setting under
    which the u8 entries and the float entry run the same arithmetic and agree bit for bit.  The default puts
    block 1 of u8 sources on the bf16 matrix pipe with three exact weight terms (other order of additions)."""
    from cutdet import api
    from cutdet.engine import Engine
    eng = Engine(api.default_blob(), torch.device("cuda:0"), "f32")
    eng.set_option("f32_block1_fp32_mfma", 1)
    yield eng
    eng.close()


# what the two block-1 formulations of the fp32 arithmetic may differ by (measured: <= 4e-6)
BLOCK1_FORMULATIONS_TOL = 2e-5


@pytest.fixture(scope="module")
def oracle_weights():
    from oracle import net as onet
    return onet.load_blob()


def _classes(dev, cl):
    return torch.tensor(cl, dtype=torch.uint8, device=dev)


# ----------------------------------------------------------------------------- generator
@pytest.mark.parametrize("h,w", [(720, 1280), (180, 320), (1080, 1920), (36, 64)])
def test_synth_frames_match_oracle(engine, h, w):
    from oracle import synth
    cl = [0, 1, 2, 1]
    got = engine.synth_frames(_classes(engine.device, cl), h, w, SEED, first_index=5).cpu().numpy()
    want = synth.clip(SEED, cl, h, w, first_index=5)
    assert got.shape == want.shape and (got == want).all()


# ----------------------------------------------------------------------------- preprocessing
@pytest.mark.parametrize("h,w", [(720, 1280), (1080, 1920), (180, 320), (480, 640), (512, 512), (288, 512), (100, 300)])
def test_preprocess_bit_exact(engine, h, w):
    from oracle import preprocess, synth
    cl = [0, 1, 2]
    frames = synth.clip(SEED, cl, h, w)
    want = preprocess.preprocess_batch(frames)
    got = engine.preprocess(torch.from_numpy(frames).to(engine.device)).cpu().numpy()
    assert got.shape == want.shape
    assert got.tobytes() == want.tobytes()


def test_preprocess_rgb_order_and_unaligned_rows(engine):
    from oracle import preprocess, synth
    frames = synth.clip(SEED, [0, 1], 90, 161)            # 483-byte rows: not 16-byte aligned
    want = preprocess.preprocess_batch(frames, chan_order=1)
    got = engine.preprocess(torch.from_numpy(frames).to(engine.device), chan_order=1).cpu().numpy()
    assert got.tobytes() == want.tobytes()
    # padded row stride (a view into a wider buffer)
    wide = torch.zeros((2, 90, 176, 3), dtype=torch.uint8, device=engine.device)
    wide[:, :, :161] = torch.from_numpy(frames).to(engine.device)
    got2 = engine.preprocess(wide[:, :, :161], chan_order=1).cpu().numpy()
    assert got2.tobytes() == want.tobytes()


# ----------------------------------------------------------------------------- network
def _uniform_input(n, h=144, w=256):
    from oracle import synth
    e = np.arange(n * 3 * h * w, dtype=np.uint32)
    with np.errstate(over="ignore"):
        r = synth.mix32(e + np.uint32(12345))
    return ((r >> np.uint32(8)).astype(np.float32) / np.float32(1 << 24)).reshape(n, 3, h, w)


def test_forward_matches_reference_logits_uniform_input(engine, golden_dir):
    g = np.load(os.path.join(golden_dir, "net_golden.npz"))
    x = torch.from_numpy(_uniform_input(4)).to(engine.device)
    y = engine.forward(x).cpu().numpy()
    assert np.abs(y - g["rand_logits"]).max() <= LOGIT_TOL


@pytest.mark.parametrize("case", ["720p", "1080p", "180p", "4x3"])
def test_forward_u8_matches_reference_logits(engine, fp32_mfma_engine, golden_dir, case):
    from oracle import synth
    g = np.load(os.path.join(golden_dir, "net_golden.npz"))
    with open(os.path.join(golden_dir, "net_golden_meta.json")) as f:
        m = json.load(f)[case]
    frames = torch.from_numpy(synth.clip(SEED, m["classes"], m["height"], m["width"])).to(engine.device)
    logits, mx, am = engine.forward_u8(frames)
    y = logits.cpu().numpy()
    err = np.abs(y - g[f"logits_{case}"]).max()
    assert err <= LOGIT_TOL, err
    assert am.cpu().tolist() == m["classes"]
    assert (mx.cpu().numpy() == y.max(axis=1)).all()
    # block 1 on the fp32 matrix instructions: also inside the contract, a few ulp from the default
    ys = fp32_mfma_engine.forward_u8(frames)[0].cpu().numpy()
    assert np.abs(ys - g[f"logits_{case}"]).max() <= LOGIT_TOL
    assert np.abs(ys - y).max() <= BLOCK1_FORMULATIONS_TOL
    # and then the two-kernel route (preprocess, then forward) gives the same bits as the fused route
    y2 = fp32_mfma_engine.forward(fp32_mfma_engine.preprocess(frames)).cpu().numpy()
    assert y2.tobytes() == ys.tobytes()
    y3 = engine.forward(engine.preprocess(frames)).cpu().numpy()             # the float entry ignores the option
    assert y3.tobytes() == y2.tobytes()


@pytest.mark.parametrize("h,w,pad", [(90, 161, 0), (180, 320, 5), (1080, 1920, 16), (100, 300, 0)])
def test_fused_general_resize_row_layouts(engine, fp32_mfma_engine, h, w, pad):
    """The fused general resize has two staging routes: source rows through LDS (16-byte aligned
    rows) and per-thread gathers (anything else: odd widths, views at odd offsets).  Both must give
    the bits of the two-kernel route, whose resize is checked byte for byte against the oracle - with
    block 1 on the fp32 matrix instructions; the default block 1 (three-term weights on the bf16 pipe) has
    the same two routes and must give the same bits on either."""
    from oracle import synth
    frames = torch.from_numpy(synth.clip(SEED, [0, 1, 2, 1, 0], h, w)).to(engine.device)
    want = fp32_mfma_engine.forward(fp32_mfma_engine.preprocess(frames))
    want_default = engine.forward_u8(frames)[0]
    if pad:
        wide = torch.zeros((5, h, w + pad + 1, 3), dtype=torch.uint8, device=engine.device)
        view = wide[:, :, 1:w + 1] if pad % 16 else wide[:, :, :w]      # odd pads: rows start off 16-byte boundaries
        view.copy_(frames)
        frames = view
    assert torch.equal(fp32_mfma_engine.forward_u8(frames)[0], want)
    assert torch.equal(engine.forward_u8(frames)[0], want_default)
    assert float((want_default - want).abs().max()) <= BLOCK1_FORMULATIONS_TOL


def test_forward_matches_oracle_mixed_batch(engine, oracle_weights):
    from oracle import net as onet
    x = _uniform_input(37)                               # odd batch size
    x[5] = 0.0
    x[6] = 1.0
    want = onet.forward(oracle_weights, x).numpy()
    logits, mx, am = engine.forward(torch.from_numpy(x).to(engine.device), want_max=True)
    got = logits.cpu().numpy()
    assert np.abs(got - want).max() <= LOGIT_TOL
    assert (am.cpu().numpy() == want.argmax(1)).all()


def test_forward_rejects_bad_shapes(engine):
    from cutdet._lib import CutdetError
    with pytest.raises(CutdetError):
        engine.forward(torch.zeros((1, 3, 144, 128), device=engine.device))
    with pytest.raises(CutdetError):
        engine.forward(torch.zeros((1, 3, 20, 256), device=engine.device))
    with pytest.raises(RuntimeError):
        engine.forward(torch.zeros((1, 3, 144, 256)))      # CPU tensor: no fallback


# ----------------------------------------------------------------------------- run-length encoding
def _seg_golden(golden_dir):
    with open(os.path.join(golden_dir, "seg_golden.json")) as f:
        cases = json.load(f)
    return cases, np.load(os.path.join(golden_dir, "seg_golden.npz"))


def _ref_means(tab):
    return np.array([np.frombuffer(bytes.fromhex(h), dtype=np.float32)[0] for h in tab["score_means_f32_hex"]],
                    dtype=np.float32)


RLE_CASES = ["ten_runs", "equal_neighbours_tie", "prev_larger", "first_orphan", "last_orphan", "tie_rows",
             "many_orphans", "chain_merge", "single_frame", "single_short_segment", "all_orphans",
             "custom_thresholds", "long_clip",
             "close_pair_p1ulp", "close_pair_m1ulp", "close_pair_p2ulp", "close_pair_m2ulp", "close_pair_p8ulp",
             "close_pair_m8ulp", "close_triple_1ulp", "close_triple_2ulp", "close_triple_mixed"]


@pytest.mark.parametrize("name", RLE_CASES)
def test_rle_matches_reference_tables(engine, golden_dir, name):
    cases, arrays = _seg_golden(golden_dir)
    c = cases[name]
    scores = torch.from_numpy(arrays[name]).to(engine.device)
    mx, am = torch.max(scores, dim=1)
    te = engine.rle(am, mx)
    for k in ("end_frames", "frame_types", "run_lengths", "start_frames"):
        assert te[k].dtype == torch.int64
        assert te[k].cpu().tolist() == c["init"][k], k
    # the per-segment means are the reference's fp32 torch.mean values bit for bit
    assert te["score_means"].cpu().numpy().tobytes() == _ref_means(c["init"]).tobytes()


def test_rle_offset_capacity_and_large_stream(engine):
    from oracle import segmentation
    rs = np.random.RandomState(3)
    n = 200_000
    runs = []
    while sum(runs) < n:
        runs.append(int(rs.randint(1, 400)))
    cls = np.concatenate([np.full(r, i % 3, np.int64) for i, r in enumerate(runs)])[:n]
    scores = rs.uniform(-2, 0, size=(n, 3)).astype(np.float32)
    scores[np.arange(n), cls] = rs.uniform(1, 9, size=n).astype(np.float32)
    want = segmentation.segments_from_scores(scores)
    st = torch.from_numpy(scores).to(engine.device)
    mx, am = torch.max(st, dim=1)
    te = engine.rle(am, mx, frame_offset=1000)
    assert te["end_frames"].cpu().tolist() == [e + 1000 for e in want["end_frames"]]
    assert te["start_frames"].cpu().tolist() == [s + 1000 for s in want["start_frames"]]
    assert te["run_lengths"].cpu().tolist() == want["run_lengths"]
    assert te["frame_types"].cpu().tolist() == want["frame_types"]
    assert te["score_means"].cpu().numpy().tobytes() == np.array(want["score_means"], dtype=np.float32).tobytes()
    # size-independent property: run lengths tile the stream exactly
    assert int(te["run_lengths"].sum().item()) == n
    from cutdet._lib import CutdetError
    with pytest.raises(CutdetError):
        engine.rle(am, mx, cap=10)
    with pytest.raises(IndexError):
        engine.rle(am[:0], mx[:0])


def test_rle_worst_case_every_frame_a_segment(engine):
    n = 10_001
    cls = (torch.arange(n, device=engine.device) % 3).to(torch.int64)
    mx = torch.arange(n, device=engine.device, dtype=torch.float32)
    te = engine.rle(cls, mx)
    assert te["end_frames"].cpu().tolist() == list(range(n))
    assert te["run_lengths"].cpu().unique().tolist() == [1]
    assert (te["score_means"].cpu() == mx.cpu()).all()


# ----------------------------------------------------------------------------- end to end
@pytest.mark.parametrize("h,w", [(180, 320), (720, 1280)])
def test_clip_end_to_end_csv_matches_oracle(engine, oracle_weights, tmp_path, h, w):
    """A small clip through the product API vs the oracle chain; CSV bytes must be identical."""
    from cutdet import synth as csynth
    from frameID.segmentation import Segmentation
    from oracle import net as onet
    from oracle import preprocess, segmentation, synth
    runs = [(0, 130), (2, 4), (1, 120), (2, 15), (0, 30), (1, 110)]
    cl = csynth.classes_from_runs(runs)
    frames_np = synth.clip(SEED, cl, h, w)
    scores = onet.forward(oracle_weights, preprocess.preprocess_batch(frames_np)).numpy()
    _, _, merged, csv = segmentation.segment_clip(scores)

    frames = engine.synth_frames(_classes(engine.device, cl), h, w, SEED)
    outs = [engine.forward_u8(frames[a:a + 128])[0] for a in range(0, len(cl), 128)]
    yy = torch.cat(outs, 0)
    assert np.abs(yy.cpu().numpy() - scores).max() <= LOGIT_TOL
    seg = Segmentation(yy)
    seg.glue_orphans(100, 10)
    seg.combine_adjacent_segments()
    for k in ("end_frames", "frame_types", "run_lengths", "start_frames"):
        assert seg.te[k].tolist() == merged[k]
    p = tmp_path / "out.csv"
    seg.write_csv(str(p))
    assert p.read_bytes() == csv


# ----------------------------------------------------------------------------- drop-in API / CLI
def _write_clip(tmp_path, runs, h, w):
    from cutdet import rawvideo, synth as csynth
    from oracle import synth
    cl = csynth.classes_from_runs(runs)
    frames = synth.clip(SEED, cl, h, w)
    p = str(tmp_path / "clip.cdraw")
    rawvideo.write_raw_video(p, frames, fps=30)
    return p, frames, cl


def test_segment_video_cli_1k_frames_320x180(engine, oracle_weights, tmp_path):
    """The reference's own CPU-runnable configuration: segment_video.py on a 1k-frame 320x180
    clip.  The CLI's CSV must equal the oracle chain's CSV byte for byte."""
    import importlib.util
    from oracle import net as onet
    from oracle import preprocess, segmentation
    runs = [(0, 300), (2, 20), (1, 250), (2, 4), (1, 126), (0, 40), (1, 160), (2, 100)]
    path, frames, cl = _write_clip(tmp_path, runs, 180, 320)
    assert len(cl) == 1000
    scores = np.concatenate([onet.forward(oracle_weights, preprocess.preprocess_batch(frames[a:a + 125])).numpy()
                             for a in range(0, 1000, 125)])
    _, _, _, want_csv = segmentation.segment_clip(scores)

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cd_segment_video",
                                                  os.path.join(root, "cut-detection_amd", "segment_video.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(mod.sv_parser.parse_args([path, "--batch-size", "128"]))
    assert out == os.path.splitext(path)[0] + "_segments.csv"
    assert open(out, "rb").read() == want_csv
    # --frame-limit keeps the reference's overshoot rule: limit 300, batch 128 -> 384 frames scored
    out2 = mod.main(mod.sv_parser.parse_args([path, "--output_path", str(tmp_path / "lim.csv"),
                                              "--frame-limit", "300", "--batch-size", "128"]))
    _, _, _, want2 = segmentation.segment_clip(scores[:384])
    assert open(out2, "rb").read() == want2
    with pytest.raises(RuntimeError):
        mod.main(mod.sv_parser.parse_args([path, "--cpu"]))


def test_sampled_row_ingest_is_bit_identical(engine, oracle_weights, tmp_path):
    """720p clip from the raw container: uploading only the 144 sampled rows of each frame and
    classifying them with net_h=144 must give the same logits, bit for bit, as the full frames -
    through the overlapped (pinned double buffer + copy stream) path and the synchronous one - and
    those logits must be the oracle's."""
    from frameID.data import VideoDataset
    from frameID.net import load_default_net
    from oracle import net as onet
    from oracle import preprocess
    runs = [(0, 20), (1, 25), (2, 19)]
    path, frames, cl = _write_clip(tmp_path, runs, 720, 1280)
    want = np.concatenate([onet.forward(oracle_weights, preprocess.preprocess_batch(frames[a:a + 16])).numpy()
                           for a in range(0, len(cl), 16)])
    net, _ = load_default_net()
    net.eval().to(engine.device)
    full = VideoDataset(path, resize=256, device=engine.device)
    a = torch.cat([net.classify_frames(b, chan_order=full.chan_order)[0] for b in full.frame_batches(32, sampled=False)])
    assert np.abs(a.cpu().numpy() - want).max() <= LOGIT_TOL
    for overlap in (True, False):
        ds = VideoDataset(path, resize=256, device=engine.device)
        assert ds.sampled_rows == [5 * y + 2 for y in range(144)]
        got = []
        for b in ds.frame_batches(24, overlap=overlap):          # 64 frames: 24 + 24 + 16, three buffer reuses
            assert tuple(b.shape[1:]) == (144, 1280, 3) and ds.batch_net_height == 144
            got.append(net.classify_frames(b, chan_order=ds.chan_order, net_h=ds.batch_net_height))
        assert torch.equal(a, torch.cat([g[0] for g in got]))
        assert torch.cat([g[2] for g in got]).cpu().tolist() == cl
    # a geometry that is not a point sample keeps whole frames
    p2, _, _ = _write_clip(tmp_path, [(0, 3)], 180, 320)
    d2 = VideoDataset(p2, resize=256, device=engine.device)
    assert d2.sampled_rows is None and next(iter(d2.frame_batches(8))).shape[1] == 180 and d2.batch_net_height is None


class _ReadOnlyCapture:
    """What cv2.VideoCapture offers the dataset: read() one BGR frame at a time, nothing else."""

    def __init__(self, path):
        from cutdet import rawvideo
        self._cap = rawvideo.RawVideoCapture(path)
        self.reads = 0

    def read(self):
        self.reads += 1
        return self._cap.read()


def test_frames_decoded_one_by_one_are_cut_to_sampled_rows_before_upload(engine, tmp_path):
    """The cv2 route (reference frameID/data.py:211-216): frames arrive one at a time from read();
    only their sampled rows are staged and uploaded, and the logits equal the whole-frame ones."""
    from frameID.data import VideoDataset
    from frameID.net import load_default_net
    path, frames, cl = _write_clip(tmp_path, [(1, 9), (0, 10)], 720, 1280)
    net, _ = load_default_net()
    net.eval().to(engine.device)
    want = net.classify_frames(torch.from_numpy(frames).to(engine.device))[0]
    ds = VideoDataset(path, resize=256, device=engine.device)
    ds.cap = _ReadOnlyCapture(path)                               # no read_batch / read_batch_into
    got = []
    for b in ds.frame_batches(8):
        assert tuple(b.shape[1:]) == (144, 1280, 3)
        got.append(net.classify_frames(b, chan_order=0, net_h=ds.batch_net_height)[0])
    assert torch.equal(torch.cat(got), want)
    assert ds.cap.reads == len(cl) + 1                            # every frame once, plus the end-of-stream read


def test_reference_style_driver_loop(engine, oracle_weights, tmp_path):
    """The reference's loop verbatim in shape: VideoDataset -> DataLoader -> net(batch) -> Segmentation."""
    from torch.utils.data import DataLoader
    from frameID.data import VideoDataset
    from frameID.net import load_default_net
    from frameID.segmentation import Segmentation
    from oracle import net as onet
    from oracle import preprocess, segmentation
    runs = [(1, 120), (2, 12), (0, 140)]
    path, frames, cl = _write_clip(tmp_path, runs, 180, 320)
    ds = VideoDataset(path, resize=256)
    assert len(ds) == len(cl) and (ds.new_height, ds.new_width) == (144, 256)
    dl = DataLoader(ds, 50)
    net, params = load_default_net()
    net.eval()
    net.to("cuda:0")
    yy = []
    with torch.no_grad():
        for batch in dl:
            assert batch.dtype == torch.float32 and tuple(batch.shape[1:]) == (3, 144, 256)
            batch = batch.to("cuda:0")
            yy.append(net(batch))
    yy = torch.cat(yy, 0).to("cpu")
    want = onet.forward(oracle_weights, preprocess.preprocess_batch(frames)).numpy()
    assert yy.shape == (len(cl), 3) and np.abs(yy.numpy() - want).max() <= LOGIT_TOL
    seg = Segmentation(yy)
    seg.glue_orphans(100, 10)
    seg.combine_adjacent_segments()
    _, _, merged, csv = segmentation.segment_clip(want)
    assert seg.te["start_frames"].tolist() == merged["start_frames"]
    assert seg.te["frame_types"].tolist() == merged["frame_types"]
    assert {k: v.dtype for k, v in seg.te.items()} == {
        "end_frames": torch.int64, "frame_types": torch.int64, "run_lengths": torch.int64,
        "start_frames": torch.int64, "score_means": torch.float32}
    p = tmp_path / "o.csv"
    seg.write_csv(str(p))
    assert p.read_bytes() == csv


def test_segmentation_class_error_cases(engine, golden_dir):
    from frameID.segmentation import Segmentation
    cases, arrays = _seg_golden(golden_dir)
    with pytest.raises(IndexError):
        Segmentation(torch.zeros((0, 3)))
    s = Segmentation(torch.from_numpy(arrays["single_short_segment"]))
    with pytest.raises(IndexError):
        s.glue_orphans(100, 10)
    for name in ("many_orphans", "long_clip", "tie_rows"):
        c = cases[name]
        s = Segmentation(torch.from_numpy(arrays[name]))
        s.glue_orphans(*c["thresholds"])
        s.combine_adjacent_segments()
        assert s.te["start_frames"].tolist() == c["merged"]["start_frames"]
        assert s.te["frame_types"].tolist() == c["merged"]["frame_types"]
        assert s.te["score_means"].numpy().tobytes() == _ref_means(c["merged"]).tobytes()


@pytest.mark.parametrize("name", [n for n in RLE_CASES if n.startswith("close_")])
def test_close_orphan_means_glue_in_the_reference_order(engine, golden_dir, name, tmp_path):
    """Adjacent orphans whose score means differ by 1, 2 or 8 float32 steps: which one is glued first
    (the smallest mean, segmentation.py:104-107) moves the segment boundary in the CSV.  The whole
    chain - GPU max/argmax, run-length pass with the reference's fp32 mean, host glue and merge -
    must give the reference's tables bit for bit and its CSV byte for byte."""
    import base64
    from frameID.segmentation import Segmentation
    cases, arrays = _seg_golden(golden_dir)
    c = cases[name]
    s = Segmentation(torch.from_numpy(arrays[name]).to(engine.device))
    for stage, step in (("init", None), ("glued", lambda: s.glue_orphans(*c["thresholds"])),
                        ("merged", s.combine_adjacent_segments)):
        if step:
            step()
        for k in ("end_frames", "frame_types", "run_lengths", "start_frames"):
            assert s.te[k].tolist() == c[stage][k], (stage, k)
        assert s.te["score_means"].numpy().tobytes() == _ref_means(c[stage]).tobytes(), stage
    p = tmp_path / "o.csv"
    s.write_csv(str(p))
    assert p.read_bytes() == base64.b64decode(c["csv_b64"])


def test_segment_means_follow_torch_mean_at_every_length(engine):
    """Every segment length from 1 to 300, lengths around the vector and cascade boundaries of
    torch's sum kernel, and the longest length that still takes that path: the GPU's per-segment mean
    equals the oracle's restatement of torch.mean (itself checked against torch) bit for bit.  From
    32768 frames on the mean is the fp64 sum rounded once."""
    from oracle import segmentation
    rs = np.random.RandomState(11)
    lengths = list(range(1, 301)) + [511, 512, 513, 1023, 1024, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 16385, 32767,
                                      32768, 50_000, 300_001]
    rs.shuffle(lengths)
    cls = np.concatenate([np.full(n, i % 3, np.int64) for i, n in enumerate(lengths)])
    mx = rs.uniform(0.5, 9, size=cls.shape[0]).astype(np.float32)
    te = engine.rle(torch.from_numpy(cls).to(engine.device), torch.from_numpy(mx).to(engine.device), frame_offset=17)
    assert te["run_lengths"].cpu().tolist() == lengths
    got = te["score_means"].cpu().numpy()
    sums = te["score_sums"].cpu().numpy()
    a = 0
    for i, n in enumerate(lengths):
        seg = mx[a:a + n]
        a += n
        exact = seg.astype(np.float64).sum()
        assert abs(sums[i] - exact) <= 1e-12 * exact
        if n < 32768:
            assert got[i] == segmentation.mean_f32_torch_order(seg), n
        else:
            assert got[i] == np.float32(sums[i] / n), n


def test_full_size_properties_10k_frames_720p(engine):
    """Headline-size run (10 000 frames, batch 256, 720p) checked through size-independent properties: the class
    stream must equal the generator's labels, segment run lengths must tile the clip, and a second
    pass over the same frames must reproduce every logit bit (the kernels are deterministic)."""
    from cutdet import synth as csynth
    from cutdet.pipeline import ClipClassifier
    n = 10_000                      # the headline configuration: 27.6 GB of frames resident in HBM
    runs = csynth.run_table(n, SEED)
    cl = csynth.classes_from_runs(runs)
    frames = engine.synth_frames(_classes(engine.device, cl), 720, 1280, SEED)
    clip = ClipClassifier(engine, n)
    for a in range(0, n, 256):
        clip.classify(frames[a:a + 256], a)
    first = clip.logits.clone()
    assert clip.cls.cpu().tolist() == cl
    for a in range(0, n, 256):
        clip.classify(frames[a:a + 256], a)
    assert torch.equal(first, clip.logits)
    res = clip.segment()
    assert int(res["initial"]["run_lengths"].sum()) == n
    assert res["initial"]["frame_types"].tolist() == [c for c, _ in runs]
    assert res["initial"]["run_lengths"].tolist() == [k for _, k in runs]
    assert int(res["merged"]["run_lengths"].sum()) == n and res["csv"].startswith(b"0,a22\r\n")


def test_stage_timing_masks_and_sampling(engine):
    """cd_timing_enable / cd_timing_interval: only the selected stages are bracketed, only every n-th
    launch of them, and reading forgets what was read."""
    frames = engine.synth_frames(_classes(engine.device, [0, 1, 2, 0]), 180, 320, SEED)
    engine.timing_read()
    engine.timing_enable(True, stages=["conv2"], every=2)
    for _ in range(4):
        engine.forward_u8(frames)
    t = engine.timing_read()
    assert t["conv2"][0] == 2 and t["conv2"][1] > 0 and t["conv1"][0] == 0 and t["conv3_head"][0] == 0
    engine.timing_enable(True)
    engine.forward_u8(frames)
    t = engine.timing_read()
    assert [t[k][0] for k in ("conv1", "conv2", "conv3_head")] == [1, 1, 1]
    engine.timing_enable(False)
    engine.forward_u8(frames)
    assert all(v[0] == 0 for v in engine.timing_read().values())


def test_fused_forward_is_hip_graph_capturable(engine):
    """After one warm-up call (workspace + resize tables in place) the fused forward queues three
    kernels and nothing else, so it can be captured into a HIP graph and replayed on new frame bytes."""
    cl = [0, 1, 2] * 8
    a = engine.synth_frames(_classes(engine.device, cl), 720, 1280, SEED)
    b = engine.synth_frames(_classes(engine.device, cl[::-1]), 720, 1280, SEED + 1)
    buf = a.clone()
    out = (torch.empty((24, 3), device=engine.device), torch.empty((24,), device=engine.device),
           torch.empty((24,), dtype=torch.int64, device=engine.device))
    want_a = [t.clone() for t in engine.forward_u8(a)]
    want_b = [t.clone() for t in engine.forward_u8(b)]
    engine.forward_u8(buf, out=out)                      # warm-up on the capture buffers
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        engine.forward_u8(buf, out=out)
    for frames, want in ((b, want_b), (a, want_a)):
        buf.copy_(frames)
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(x, y) for x, y in zip(out, want))


# ----------------------------------------------------------------------------- stream discipline
def test_reserve_then_forward_on_a_side_stream(oracle_weights):
    """cd_reserve() zeroes the workspace borders; a forward on another (non-blocking) stream right
    after it must see them zeroed, i.e. give the default-stream result."""
    from cutdet import api
    from cutdet.engine import Engine
    dev = torch.device("cuda", 0)
    ref_eng = api.default_engine(dev)
    cl = [0, 1, 2, 1, 0, 2, 1, 0] * 8
    frames = ref_eng.synth_frames(_classes(dev, cl), 720, 1280, SEED, first_index=11)
    want = ref_eng.forward_u8(frames)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    for h in (144, 85):
        eng = Engine(api.default_blob(), dev)
        # dirty memory first, so that stale borders would show
        junk = torch.full((int(3e8 // 4),), float("nan"), device=dev)
        del junk
        eng.reserve(len(cl), h)
        with torch.cuda.stream(side):
            got = eng.forward_u8(frames, net_h=h)
        side.synchronize()
        if h == 144:
            for w, g in zip(want, got):
                assert torch.equal(w, g)
        else:
            assert torch.isfinite(got[0]).all()
        eng.close()
