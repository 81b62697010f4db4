"""ctypes binding of libcutdet.so (declarations: include/cutdet.h).

There is no CPU implementation behind these calls: if the library is missing the import of
this module raises, and every call needs device memory on an MI355X.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CUTDET_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libcutdet.so"))

if not os.path.isfile(LIB_PATH):
    raise ImportError(
        f"cutdet: native library not found at {LIB_PATH}; build it with "
        f"`make -C {os.path.join(os.path.dirname(_HERE), 'csrc')}` (there is no CPU fallback)")

lib = C.CDLL(LIB_PATH)

CD_OK, CD_EINVAL, CD_EHIP, CD_ECAP, CD_ENOMEM = 0, -1, -2, -3, -4
CD_F32 = 0
CD_BF16 = 1
CD_BF16_SPLIT = 2
DTYPES = {"f32": CD_F32, "bf16": CD_BF16, "bf16_split": CD_BF16_SPLIT}
CD_BGR, CD_RGB = 0, 1
OPTIONS = {"range_by_stage": 0, "fuse_blocks12": 1, "range_one_stream": 2, "split_two_waves": 3, "resize_rows": 4, "f32_block1_fp32_mfma": 5}
STAGES = ("preprocess", "conv1", "conv2", "conv3_head", "rle")

_h = C.c_void_p
_p = C.c_void_p
_i = C.c_int
_l = C.c_int64

SIGNATURES = {
    "cd_version": (C.c_int, []),
    "cd_last_error": (C.c_char_p, []),
    "cd_create": (C.c_int, [C.POINTER(_h), _p, C.c_size_t, _i, _i]),
    "cd_destroy": (None, [_h]),
    "cd_reserve": (C.c_int, [_h, _i, _i]),
    "cd_preprocess": (C.c_int, [_h, _p, _i, _i, _i, _l, _l, _i, _p, _i, _i, _p]),
    "cd_resize_u8": (C.c_int, [_h, _p, _i, _i, _i, _l, _l, _p, _i, _i, _p]),
    "cd_nv12_to_bgr": (C.c_int, [_h, _p, _p, _i, _i, _l, _l, _l, _l, _p, _p, _i, _p, _i, _p]),
    "cd_forward": (C.c_int, [_h, _p, _i, _i, _i, _p, _p, _p, _p]),
    "cd_forward_u8": (C.c_int, [_h, _p, _i, _i, _i, _l, _l, _i, _i, _i, _p, _p, _p, _p]),
    "cd_forward_u8_range": (C.c_int, [_h, _p, _l, _i, _i, _i, _l, _l, _i, _i, _i, _p, _p, _p, _p]),
    "cd_set_option": (C.c_int, [_h, _i, _i]),
    "cd_forward_u8_static": (C.c_int, [_h, _p, _i, _i, _i, _l, _l, _i, _i, _i, _p, _p, _p, _p]),
    "cd_max_argmax": (C.c_int, [_h, _p, _l, _i, _p, _p, _p]),
    "cd_eval_accumulate": (C.c_int, [_h, _p, _p, _l, _i, _p, _p, _p, _p]),
    "cd_rle": (C.c_int, [_h, _p, _p, _l, _l, _p, _p, _p, _p, _p, _p, _l, _p, _p]),
    "cd_synth_frames": (C.c_int, [_h, _p, _i, _i, _i, _p, C.c_uint32, _l, _p]),
    "cd_timing_enable": (C.c_int, [_h, _i]),
    "cd_timing_interval": (C.c_int, [_h, _i]),
    "cd_timing_read": (C.c_int, [_h, _i, C.POINTER(_l), C.POINTER(C.c_double)]),
    "cd_glue_segments": (C.c_int, [_p, _p, _p, _p, _p, C.POINTER(_l), _l, _l, _i]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


class CutdetError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libcutdet error {code}: {message}")
        self.code = code
        self.message = message


def check(code):
    if code != CD_OK:
        raise CutdetError(code, (lib.cd_last_error() or b"").decode("utf-8", "replace"))
