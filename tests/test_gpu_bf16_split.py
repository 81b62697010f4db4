"""Split-bf16 mode (CD_BF16X3): blocks 2 and 3 on the bf16 matrix cores with every fp32 factor
split into hi + lo bfloat16 terms (three products, fp32 accumulation).

Unlike plain bf16 this mode is held to the fp32 contract: logits within 1e-3 of the logits the
REFERENCE produced (tests/golden/net_golden.npz) on all four golden geometries and on the
random-input case, class ids / segments / CSV identical to the fp32 chain."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED = 1234
LOGIT_TOL = 1e-3


@pytest.fixture(scope="module")
def engines():
    from cutdet import api
    return api.default_engine("cuda:0"), api.default_engine("cuda:0", "bf16_split")


def _uniform_input(n, h=144, w=256):
    from oracle import synth
    e = np.arange(n * 3 * h * w, dtype=np.uint32)
    with np.errstate(over="ignore"):
        r = synth.mix32(e + np.uint32(12345))
    return ((r >> np.uint32(8)).astype(np.float32) / np.float32(1 << 24)).reshape(n, 3, h, w)


@pytest.mark.parametrize("case", ["720p", "1080p", "180p", "4x3"])
def test_x3_matches_reference_logits(engines, golden_dir, case):
    from oracle import synth
    e32, ex = engines
    g = np.load(os.path.join(golden_dir, "net_golden.npz"))
    with open(os.path.join(golden_dir, "net_golden_meta.json")) as f:
        m = json.load(f)[case]
    frames = torch.from_numpy(synth.clip(SEED, m["classes"], m["height"], m["width"])).to(ex.device)
    logits, mx, am = ex.forward_u8(frames)
    y = logits.cpu().numpy()
    err = float(np.abs(y - g[f"logits_{case}"]).max())
    print(f"bf16_split {case}: max |logit - reference| = {err:.2e}; vs fp32 kernels "
          f"{float((logits - e32.forward_u8(frames)[0]).abs().max()):.2e}")
    assert err <= LOGIT_TOL, err
    assert am.cpu().tolist() == m["classes"]
    assert (mx.cpu().numpy() == y.max(axis=1)).all()


def test_x3_random_input_and_odd_heights(engines, golden_dir):
    from oracle import net as onet
    e32, ex = engines
    g = np.load(os.path.join(golden_dir, "net_golden.npz"))
    x = torch.from_numpy(_uniform_input(4)).to(ex.device)
    assert np.abs(ex.forward(x).cpu().numpy() - g["rand_logits"]).max() <= LOGIT_TOL
    w = onet.load_blob()
    gen = torch.Generator().manual_seed(7)
    for h in (27, 100, 144, 192, 215):
        x = torch.rand((5, 3, h, 256), generator=gen)
        got = ex.forward(x.to(ex.device)).cpu()
        assert float((got - onet.forward(w, x)).abs().max()) <= LOGIT_TOL


def test_x3_clip_segments_identical_to_fp32_chain(engines):
    from cutdet import synth as csynth
    from cutdet.pipeline import ClipClassifier
    e32, ex = engines
    n = 2048
    cl = csynth.classes_from_runs(csynth.run_table(n, SEED))
    frames = e32.synth_frames(torch.tensor(cl, dtype=torch.uint8, device=e32.device), 720, 1280, SEED)
    out = []
    for eng in (e32, ex):
        clip = ClipClassifier(eng, n)
        for a in range(0, n, 256):
            clip.classify(frames[a:a + 256], a)
        out.append((clip.cls.cpu().tolist(), clip.logits.cpu(), clip.segment()))
    (c32, l32, s32), (cx, lx, sx) = out
    assert cx == c32 == cl
    assert float((lx - l32).abs().max()) <= LOGIT_TOL
    assert sx["csv"] == s32["csv"]
    for k in ("end_frames", "frame_types", "run_lengths", "start_frames"):
        assert sx["merged"][k].tolist() == s32["merged"][k].tolist()
