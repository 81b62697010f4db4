"""The oracle against the fixtures captured from the reference (CPU only)."""
import base64
import json
import os

import numpy as np
import torch
import pytest

from oracle import net as onet
from oracle import preprocess, segmentation, synth

SEED = 1234


def _uniform_input(n, h=144, w=256):
    e = np.arange(n * 3 * h * w, dtype=np.uint32)
    with np.errstate(over="ignore"):
        r = synth.mix32(e + np.uint32(12345))
    return ((r >> np.uint32(8)).astype(np.float32) / np.float32(1 << 24)).reshape(n, 3, h, w)


@pytest.fixture(scope="module")
def weights():
    return onet.load_blob()


@pytest.fixture(scope="module")
def net_golden(golden_dir):
    with open(os.path.join(golden_dir, "net_golden_meta.json")) as f:
        meta = json.load(f)
    return np.load(os.path.join(golden_dir, "net_golden.npz")), meta


@pytest.mark.parametrize("case", ["720p", "1080p", "180p", "4x3"])
def test_net_logits_match_reference(weights, net_golden, case):
    g, meta = net_golden
    m = meta[case]
    frames = synth.clip(SEED, m["classes"], m["height"], m["width"])
    x = preprocess.preprocess_batch(frames)
    assert list(x.shape) == m["input_shape"]
    y = onet.forward(weights, x).numpy()
    # same op sequence as the reference; oneDNN picks its conv primitive per process, which
    # moves fp32 logits by a few 1e-6 between runs of the SAME code, so allow 2e-5
    np.testing.assert_allclose(y, g[f"logits_{case}"], rtol=0, atol=2e-5)
    assert (y.argmax(1) == np.array(m["classes"])).all()


def test_net_layers_match_reference(weights, net_golden):
    g, _ = net_golden
    x = preprocess.preprocess_batch(synth.clip(SEED, [0, 1, 2], 720, 1280))
    a1, a2, a3, feat, hid, logits = [t.numpy() for t in onet.forward_layers(weights, x)]
    for i, a in enumerate((a1, a2, a3)):
        np.testing.assert_allclose(a.astype(np.float64).sum(axis=(2, 3)), g[f"act{i + 1}_chansum"], rtol=1e-5, atol=2e-3)
        np.testing.assert_allclose(a[:, :, :5, :8], g[f"act{i + 1}_tl"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(a[:, :, -5:, -8:], g[f"act{i + 1}_br"], rtol=0, atol=2e-5)
    assert a1.shape[1:] == (48, 48, 85) and a2.shape[1:] == (48, 16, 28) and a3.shape[1:] == (48, 5, 9)
    np.testing.assert_allclose(feat, g["feat"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(hid, g["hidden"], rtol=0, atol=2e-5)
    assert float(g["traced_minus_eager_maxabs"]) == 0.0


def test_net_uniform_input(weights, net_golden):
    g, _ = net_golden
    y = onet.forward(weights, _uniform_input(4)).numpy()
    np.testing.assert_allclose(y, g["rand_logits"], rtol=0, atol=2e-5)


def test_traced_equals_eager(weights):
    import torch
    x = torch.from_numpy(_uniform_input(2))
    t = onet.traced_cpu_net(weights)
    assert (t(x) - onet.forward(weights, x)).abs().max().item() == 0.0


def test_bf16_restatement_stays_near_the_pinned_fp32_net(weights):
    """forward_bf16 (what the GPU's bf16 mode is checked against) rounds inputs, weights and the first
    two activations to bfloat16; it must stay within a few 1e-2 of the pinned fp32 forward and keep
    the class of every synthetic generator."""
    frames = np.stack([synth.frame(SEED, i, i % 3, 720, 1280) for i in range(6)])
    x = preprocess.preprocess_batch(frames)
    y32, y16 = onet.forward(weights, x), onet.forward_bf16(weights, x)
    assert 1e-4 < float((y32 - y16).abs().max()) <= 8e-2
    assert y16.argmax(1).tolist() == y32.argmax(1).tolist() == [0, 1, 2, 0, 1, 2]


# ----------------------------------------------------------------------------- segmentation
def _load_seg(golden_dir):
    with open(os.path.join(golden_dir, "seg_golden.json")) as f:
        cases = json.load(f)
    arrays = np.load(os.path.join(golden_dir, "seg_golden.npz"))
    return cases, arrays


def _check_table(te, ref, exact_means):
    for k in ("end_frames", "frame_types", "run_lengths", "start_frames"):
        assert [int(v) for v in te[k]] == ref[k], k
    want = np.array([np.frombuffer(bytes.fromhex(h), dtype=np.float32)[0] for h in ref["score_means_f32_hex"]],
                    dtype=np.float32)
    got = np.array(te["score_means"], dtype=np.float32)
    if exact_means:
        assert got.tobytes() == want.tobytes()
    else:
        np.testing.assert_allclose(got, want, rtol=1e-6)


SEG_CASES = ["ten_runs", "equal_neighbours_tie", "prev_larger", "first_orphan", "last_orphan", "tie_rows",
             "many_orphans", "chain_merge", "single_frame", "single_short_segment", "all_orphans", "empty",
             "custom_thresholds", "long_clip",
             # two / three adjacent orphans whose score means are 1, 2 and 8 float32 steps apart: the order
             # in which they are glued, and with it the CSV, hangs on the exact fp32 mean of the reference
             "close_pair_p1ulp", "close_pair_m1ulp", "close_pair_p2ulp", "close_pair_m2ulp", "close_pair_p8ulp",
             "close_pair_m8ulp", "close_triple_1ulp", "close_triple_2ulp", "close_triple_mixed"]


@pytest.mark.parametrize("name", SEG_CASES)
def test_segmentation_matches_reference(golden_dir, name):
    cases, arrays = _load_seg(golden_dir)
    c = cases[name]
    scores = arrays[name]
    k1, kb = c["thresholds"]
    if "init" not in c:
        with pytest.raises(IndexError):
            segmentation.segments_from_scores(scores)
        return
    te = segmentation.segments_from_scores(scores)
    _check_table(te, c["init"], exact_means=True)
    if "merged" not in c:
        assert c["raises"] == "IndexError"
        with pytest.raises(IndexError):
            segmentation.glue_orphans(te, k1, kb)
        return
    segmentation.glue_orphans(te, k1, kb)
    _check_table(te, c["glued"], exact_means=True)
    segmentation.combine_adjacent(te)
    _check_table(te, c["merged"], exact_means=True)
    assert segmentation.csv_bytes(te) == base64.b64decode(c["csv_b64"])


def test_known_csv_example(golden_dir):
    cases, _ = _load_seg(golden_dir)
    csv = base64.b64decode(cases["ten_runs"]["csv_b64"])
    assert csv.endswith(b"\r\n") and b"," in csv and csv.split(b"\r\n")[0] == b"0,a22"


# ----------------------------------------------------------------------------- preprocessing
def test_output_size():
    assert preprocess.output_size(720, 1280) == (144, 256)
    assert preprocess.output_size(1080, 1920) == (144, 256)
    assert preprocess.output_size(180, 320) == (144, 256)
    assert preprocess.output_size(480, 640) == (192, 256)


def test_720p_resize_is_point_sample():
    f = synth.frame(SEED, 3, 0, 720, 1280)
    r = preprocess.resize_linear_u8(f, 144, 256)
    assert (r == f[2::5, 2::5]).all()


def test_resize_identity_and_constant():
    f = synth.frame(SEED, 5, 0, 36, 64)
    assert (preprocess.resize_linear_u8(f, 36, 64) == f).all()
    c = np.full((90, 160, 3), 77, np.uint8)
    assert (preprocess.resize_linear_u8(c, 36, 64) == 77).all()


def test_1080p_weights_quarter_three_quarter():
    # scale 7.5: even outputs sample at +0.25, odd at +0.75 of a source pixel pair
    f = synth.frame(SEED, 9, 0, 1080, 1920).astype(np.int64)
    r = preprocess.resize_linear_u8(f.astype(np.uint8), 144, 256)
    y, x = 10, 21  # fy: (10.5*7.5-.5)=78.25 -> rows 78,79 (.75,.25); fx: 160.75 -> cols 160,161 (.25,.75)
    for c in range(3):
        s0 = f[78, 160, c] * 512 + f[78, 161, c] * 1536
        s1 = f[79, 160, c] * 512 + f[79, 161, c] * 1536
        want = (((1536 * (s0 >> 4)) >> 16) + ((512 * (s1 >> 4)) >> 16) + 2) >> 2
        assert r[y, x, c] == want


def test_tensorise_layout():
    f = synth.frame(SEED, 1, 0, 4, 6)
    t = preprocess.tensorise(f)
    assert t.shape == (3, 4, 6) and t.dtype == np.float32
    assert t[0, 1, 2] == np.float32(f[1, 2, 2]) / np.float32(255)  # R plane <- memory channel 2 (BGR)
    assert t[2, 3, 5] == np.float32(f[3, 5, 0]) / np.float32(255)


def test_synth_classes_and_determinism():
    a = synth.frame(SEED, 0, 0, 8, 16)
    assert (a == synth.frame(SEED, 0, 0, 8, 16)).all() and (a != synth.frame(SEED, 1, 0, 8, 16)).any()
    assert synth.frame(SEED, 0, 2, 8, 16).max() <= 7
    e = synth.frame(SEED, 0, 1, 720, 1280)
    assert e[0:5].min() >= 230 and e[5:60, :, 1].min() >= 128 and e[5:60, :, 0].max() <= 33


def test_mean_restatement_is_torch_mean_bit_for_bit():
    """oracle.segmentation.mean_f32_torch_order against torch.mean on this machine: every length up
    to 70, lengths around the vector / cascade boundaries, up to just below the 32768 elements from
    which torch splits the reduction over threads."""
    rs = np.random.RandomState(7)
    for n in list(range(1, 71)) + [127, 128, 129, 511, 512, 513, 1023, 2049, 4095, 4096, 8191, 8192, 8193, 20011, 32767]:
        for scale in (1.0, 1e-3):
            x = (rs.uniform(1, 9, size=n) * scale).astype(np.float32)
            assert segmentation.mean_f32_torch_order(x) == np.float32(torch.from_numpy(x).mean().item()), n


def test_close_means_cases_separate_the_fp32_mean_from_a_rounded_fp64_mean(golden_dir):
    """What the close-means fixtures are for: in some of them the order of two orphans' means under
    the reference's fp32 summation is NOT the order of the exactly rounded means, so an
    implementation that sums in fp64 and rounds once picks the other orphan first."""
    cases, arrays = _load_seg(golden_dir)
    flips = []
    for name in SEG_CASES:
        if not name.startswith("close_pair"):
            continue
        s = arrays[name]
        best = s.max(axis=1)
        a, b = best[300:330], best[330:360]
        f32 = (segmentation.mean_f32_torch_order(a), segmentation.mean_f32_torch_order(b))
        f64 = (np.float32(a.astype(np.float64).sum() / 30), np.float32(b.astype(np.float64).sum() / 30))
        assert f32[0] != f32[1]
        if (f32[0] < f32[1]) != (f64[0] < f64[1]) or f64[0] == f64[1]:
            flips.append(name)
    assert flips, "no fixture distinguishes the two summations"
