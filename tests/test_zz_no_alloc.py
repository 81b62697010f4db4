"""The allocation-free forward (cd_forward_u8_static) against cd_forward_u8, through the C ABI.

Bar: bit-identical logits, maxima and class ids (the same kernels run on another activation
buffer).  Two calls in a row, the second smaller; then another input height (the array's zero
borders move) and back.  Must pass alone in a fresh process and after the other GPU tests."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 1234


def _outs(dev, n):
    return (torch.full((n, 3), float("nan"), dtype=torch.float32, device=dev),
            torch.full((n,), float("nan"), dtype=torch.float32, device=dev),
            torch.full((n,), -1, dtype=torch.int64, device=dev))


def _call(lib, name, eng, frames, net_h, outs):
    n, H0, W0, _ = frames.shape
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    stream = C.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream)
    return getattr(lib, name)(eng._h, p(frames), n, H0, W0, frames.stride(1), frames.stride(0), 0, net_h, 256,
                              p(outs[0]), p(outs[1]), p(outs[2]), stream)


def test_static_forward_matches_forward_u8_on_consecutive_calls():
    from cutdet import _lib, api
    eng = api.default_engine("cuda:0")
    dev = eng.device
    lib = _lib.lib
    cl = [0, 1, 2, 1, 0, 2, 1, 1, 0, 0, 2, 1] * 3                       # 36 frames > the 34 that fit
    big = eng.synth_frames(torch.tensor(cl, dtype=torch.uint8, device=dev), 720, 1280, SEED)
    small = eng.synth_frames(torch.tensor(cl[:9], dtype=torch.uint8, device=dev), 180, 320, SEED, first_index=77)
    short = eng.synth_frames(torch.tensor(cl[:5], dtype=torch.uint8, device=dev), 100, 300, SEED, first_index=3)
    torch.cuda.synchronize()

    def both(frames, net_h):
        want, got = _outs(dev, frames.shape[0]), _outs(dev, frames.shape[0])
        assert _call(lib, "cd_forward_u8", eng, frames, net_h, want) == _lib.CD_OK
        free0 = torch.cuda.mem_get_info(dev)[0]
        assert _call(lib, "cd_forward_u8_static", eng, frames, net_h, got) == _lib.CD_OK, lib.cd_last_error()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free0, "the static entry point changed the free device memory"
        for w, g in zip(want, got):
            assert not torch.isnan(g.float()).any()
            assert torch.equal(w, g)
        return got

    first = both(big[:34], 144)            # the largest batch that fits at 144 rows
    assert first[2].cpu().tolist() == cl[:34]
    both(big[3:14], 144)                   # straight after, smaller: the array still holds the first call's data
    both(small, 144)                       # other source geometry, same network height
    both(short, 85)                        # other network height: the zero borders move
    again = both(big[:34], 144)            # and back
    for a, b in zip(first, again):
        assert torch.equal(a, b)

    # beyond the limit: refused with a message, nothing launched
    outs = _outs(dev, 36)
    rc = _call(lib, "cd_forward_u8_static", eng, big, 144, outs)
    assert rc == _lib.CD_ECAP
    assert b"34 frames" in lib.cd_last_error()
    assert torch.isnan(outs[0]).all()
    # the Python wrapper raises for the same case
    with pytest.raises(_lib.CutdetError):
        eng.forward_u8_static(big, _outs(dev, 36))


def test_static_forward_alternating_cell_sizes_at_one_height():
    """Handles of different arithmetics share the GPU's static array: an fp32 (or split-bf16) handle
    stores 4-byte activation cells and a bf16 handle 2-byte ones over the same [rows][cols][48]
    grid, so their zero borders lie at different addresses.  Alternating them at ONE network height
    must still reproduce each handle's cd_forward_u8 bit for bit."""
    from cutdet import _lib, api
    lib = _lib.lib
    engines = [api.default_engine("cuda:0", d) for d in ("f32", "bf16", "bf16_split", "bf16", "f32")]
    dev = engines[0].device
    cl = [0, 1, 2, 1, 0, 2, 1, 1, 0, 0, 2, 1, 0, 2]
    frames = engines[0].synth_frames(torch.tensor(cl, dtype=torch.uint8, device=dev), 720, 1280, SEED, first_index=11)
    torch.cuda.synchronize()
    for eng in engines:
        want, got = _outs(dev, len(cl)), _outs(dev, len(cl))
        assert _call(lib, "cd_forward_u8", eng, frames, 144, want) == _lib.CD_OK
        assert _call(lib, "cd_forward_u8_static", eng, frames, 144, got) == _lib.CD_OK, lib.cd_last_error()
        torch.cuda.synchronize()
        for w, g in zip(want, got):
            assert torch.equal(w, g), eng.dtype
        assert got[2].cpu().tolist() == cl
