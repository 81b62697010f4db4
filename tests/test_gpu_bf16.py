"""bf16 mode of the HIP path (the three convolutions on the bf16 matrix cores, fp32 accumulation).

The reference computes in fp32 only, so this mode has no reference fixture: "parity unpinned" beyond
what follows.  It is held to (a) the oracle's restatement of the same arithmetic
(oracle.net.forward_bf16: identical rounding points, fp32 sums in another order), within 5e-3 -
measured 3e-4..1.9e-3: a sum that lands across a bf16 rounding boundary moves one activation by a
bf16 ulp; (b) the pinned fp32 oracle within 8e-2 on logits of magnitude 5-40 (measured
3.7e-2..4.2e-2: bf16 keeps 8 mantissa bits of every input and activation); (c) class ids, segment
tables and CSV bytes identical to the fp32 chain on the synthetic clips, whose class margins
(>= 2.4) dwarf that error.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED = 1234
TOL_SAME_ARITHMETIC = 5e-3
TOL_VS_FP32 = 8e-2


@pytest.fixture(scope="module")
def engines():
    from cutdet import api
    return api.default_engine("cuda:0"), api.default_engine("cuda:0", "bf16")


@pytest.fixture(scope="module")
def oracle_weights():
    from oracle import net as onet
    return onet.load_blob()


@pytest.mark.parametrize("h,w,n", [(720, 1280, 24), (1080, 1920, 12), (180, 320, 24), (480, 640, 9)])
def test_bf16_forward_u8_matches_its_restatement_and_fp32(engines, oracle_weights, h, w, n):
    from oracle import net as onet
    e32, e16 = engines
    classes = torch.tensor([i % 3 for i in range(n)], dtype=torch.uint8, device=e32.device)
    frames = e32.synth_frames(classes, h, w, SEED)
    l16, m16, c16 = e16.forward_u8(frames)
    x = e32.preprocess(frames).cpu()          # bit-exact against the oracle's resize elsewhere in the suite
    ref16 = onet.forward_bf16(oracle_weights, x)
    ref32 = onet.forward(oracle_weights, x)
    assert float((l16.cpu() - ref16).abs().max()) <= TOL_SAME_ARITHMETIC
    assert float((l16.cpu() - ref32).abs().max()) <= TOL_VS_FP32
    assert c16.cpu().tolist() == ref32.argmax(1).tolist() == classes.cpu().tolist()
    mx, am = torch.max(l16, 1)
    assert torch.equal(mx, m16) and torch.equal(am, c16)


def test_bf16_float_entry_and_odd_heights(engines, oracle_weights):
    from oracle import net as onet
    _, e16 = engines
    g = torch.Generator().manual_seed(7)
    for h in (27, 100, 144, 215):
        x = torch.rand((5, 3, h, 256), generator=g)
        got = e16.forward(x.to(e16.device)).cpu()
        assert float((got - onet.forward_bf16(oracle_weights, x)).abs().max()) <= TOL_SAME_ARITHMETIC


def test_bf16_clip_1080p_segments_identical_to_fp32_chain(engines):
    """The 1080p bf16 configuration: logits move, class ids, segments and CSV do not."""
    from cutdet import synth as csynth
    from cutdet.pipeline import ClipClassifier
    e32, e16 = engines
    n = 1536
    runs = csynth.run_table(n, SEED)
    cl = csynth.classes_from_runs(runs)
    frames = e32.synth_frames(torch.tensor(cl, dtype=torch.uint8, device=e32.device), 1080, 1920, SEED)
    out = []
    for eng in (e32, e16):
        clip = ClipClassifier(eng, n)
        for a in range(0, n, 512):
            clip.classify(frames[a:a + 512], a)
        out.append((clip.cls.cpu().tolist(), clip.logits.cpu(), clip.segment()))
    (c32, l32, s32), (c16, l16, s16) = out
    assert c16 == c32 == cl
    assert float((l16 - l32).abs().max()) <= TOL_VS_FP32
    for k in ("end_frames", "frame_types", "run_lengths", "start_frames"):
        assert s16["merged"][k].tolist() == s32["merged"][k].tolist()
    assert s16["csv"] == s32["csv"]


def test_bf16_is_deterministic_and_handles_are_independent(engines):
    e32, e16 = engines
    frames = e32.synth_frames(torch.tensor([0, 1, 2] * 11, dtype=torch.uint8, device=e32.device), 720, 1280, 3)
    a = e16.forward_u8(frames)[0].clone()
    b32 = e32.forward_u8(frames)[0].clone()      # the other handle's workspace is untouched by this
    assert torch.equal(a, e16.forward_u8(frames)[0])
    assert torch.equal(b32, e32.forward_u8(frames)[0])
    assert not torch.equal(a, b32)


def test_create_rejects_unknown_dtype():
    from cutdet import _lib, api
    h = C.c_void_p(0)
    blob = api.default_blob()
    rc = _lib.lib.cd_create(C.byref(h), blob, len(blob), 0, 3)
    assert rc == _lib.CD_EINVAL and b"dtype" in _lib.lib.cd_last_error()
    with pytest.raises(ValueError):
        from cutdet.engine import Engine
        Engine(blob, "cuda:0", dtype="fp8")


def test_random_views_both_modes(engines, oracle_weights):
    """Batch sizes that do not fill the 8-way workgroup interleave, padded row strides, cropped
    views with an offset start, either channel order, assorted geometries: resized floats exact,
    fp32 logits within 1e-3 of the oracle, bf16 logits within the bf16 bars."""
    from cutdet import _lib
    from oracle import net as onet, preprocess as opre
    e32, e16 = engines
    rng = np.random.default_rng(11)
    geoms = [(720, 1280), (1080, 1920), (180, 320), (360, 640), (100, 300), (430, 512)]
    for trial in range(12):
        H0, W0 = geoms[trial % len(geoms)]
        n = int(rng.choice([1, 2, 3, 5, 7, 9, 17]))
        pad = int(rng.choice([0, 3, 16, 37]))
        order = trial & 1
        full = torch.from_numpy(rng.integers(0, 256, size=(n, H0 + 2, W0 + pad, 3), dtype=np.uint8)).to(e32.device)
        frames = full[:, 1:H0 + 1, :W0, :]
        x_ref = opre.preprocess_batch(frames.cpu().numpy(), chan_order=order)
        assert np.array_equal(e32.preprocess(frames, chan_order=order).cpu().numpy(), x_ref)
        ref32, ref16 = onet.forward(oracle_weights, x_ref), onet.forward_bf16(oracle_weights, x_ref)
        l32 = e32.forward_u8(frames, chan_order=order)[0].cpu()
        l16 = e16.forward_u8(frames, chan_order=order)[0].cpu()
        assert float((l32 - ref32).abs().max()) <= 1e-3, (trial, H0, W0, n, pad, order)
        assert float((l16 - ref16).abs().max()) <= TOL_SAME_ARITHMETIC, (trial, H0, W0, n, pad, order)
        assert float((l16 - ref32).abs().max()) <= 0.25      # random noise frames: larger activations than the clips


def test_random_weights_including_negative_batchnorm_scales(oracle_weights):
    """The bundled net has only positive BatchNorm scales and one set of magnitudes; a random net
    (negative scales, biased running means, all tensors random) checks the weight packing and the
    order ReLU -> max-pool -> affine in every block, in both arithmetic modes."""
    import struct
    from cutdet import weights as cw
    from cutdet.engine import Engine
    from oracle import net as onet
    rng = np.random.default_rng(5)
    parts = []
    for name, shape in cw.tensor_plan():
        leaf = name.split(".")[-1]
        if name.endswith("bn.weight"):
            a = rng.uniform(0.4, 1.5, shape) * rng.choice([-1.0, 1.0], shape)
        elif leaf == "running_var":
            a = rng.uniform(0.5, 2.0, shape)
        elif leaf == "running_mean":
            a = rng.normal(0.0, 0.3, shape)
        elif leaf == "weight" and len(shape) >= 2:
            a = rng.normal(0.0, 1.0 / np.sqrt(np.prod(shape[1:])), shape)
        else:
            a = rng.normal(0.0, 0.2, shape)
        parts.append(np.asarray(a, dtype="<f4").ravel())
    flat = np.concatenate(parts)
    blob = open(onet._BLOB, "rb").read()[:8] + struct.pack("<8I", 1, 3, 48, 3, 4, 32, 3, flat.size) + flat.tobytes()
    w = onet.load_blob_bytes(blob)
    assert (w["c1.g"] < 0).any() and (w["c0.g"] < 0).any() and (w["fc0.g"] < 0).any()
    x = torch.from_numpy(rng.random((6, 3, 144, 256), dtype=np.float32))
    frames = torch.from_numpy(rng.integers(0, 256, size=(5, 720, 1280, 3), dtype=np.uint8))
    ref32, ref16 = onet.forward(w, x), onet.forward_bf16(w, x)
    scale = float(ref32.abs().max())
    for dtype, ref, tol in (("f32", ref32, 1e-3), ("bf16", ref16, TOL_SAME_ARITHMETIC)):
        eng = Engine(blob, "cuda:0", dtype=dtype)
        got = eng.forward(x.cuda()).cpu()
        assert float((got - ref).abs().max()) <= tol * max(1.0, scale), dtype
        xu = eng.preprocess(frames.cuda())
        a, b = eng.forward(xu).cpu(), eng.forward_u8(frames.cuda())[0].cpu()
        if dtype == "f32":
            # default block 1 of u8 sources: three exact weight terms on the bf16 pipe, a few ulp from the float entry
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, scale)
            eng.set_option("f32_block1_fp32_mfma", 1)
            assert torch.equal(a, eng.forward_u8(frames.cuda())[0].cpu())   # fused resize = resize then forward, bit for bit
        else:
            assert float((a - b).abs().max()) <= tol * max(1.0, scale)   # the fused path rounds the same pixels to bf16
        eng.close()
