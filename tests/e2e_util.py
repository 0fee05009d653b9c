"""helpers for the end-to-end bitstream tests: drive the reference encoder through its C API — the plain build (oracle/_ref/libvvenc_ref.so, the CPU oracle) and
the build WITH the MI355X binding (bindings/vvenc/_build/libvvenc_hip_enc.so, product; bindings/vvenc/Makefile)"""
import ctypes as C
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libvvenc_ref.so")
REF_HIP_SO = os.path.join(ROOT, "bindings", "vvenc", "_build", "libvvenc_hip_enc.so")
PRESET_FASTER, PRESET_FAST, PRESET_MEDIUM = 0, 1, 2   # vvencPresetMode (include/vvenc/vvencCfg.h)
PRESETS = {"faster": 0, "fast": 1, "medium": 2}


def synth_yuv(width, height, frames, bit_depth, seed):
    """BASELINE config-1 style generator: moving sinusoid texture + noise, smooth chroma; returns int16 planes"""
    rng = np.random.default_rng(seed)
    maxv = (1 << bit_depth) - 1
    s = maxv / 255.0
    yy, xx = np.mgrid[0:height, 0:width]
    ys, us, vs = [], [], []
    for t in range(frames):
        y = 128 + 60 * np.sin((xx + 2 * t) / 7.0) + 40 * np.cos((yy - t) / 5.0) + rng.integers(-4, 5, size=(height, width))
        ys.append(np.clip(y * s, 0, maxv))
        cy, cx = np.mgrid[0:height // 2, 0:width // 2]
        us.append(np.clip((128 + 20 * np.sin((cx + t) / 9.0)) * s, 0, maxv))
        vs.append(np.clip((128 + 20 * np.cos((cy - t) / 11.0)) * s, 0, maxv))
    f = lambda a: np.ascontiguousarray(np.stack(a).astype(np.int16))
    return f(ys), f(us), f(vs)


def load(hip=False, path=None):
    L = C.CDLL(path or (REF_HIP_SO if hip else REF_SO))
    if hip:
        L.vvref_encode_ex = L.vvenc_hip_encode          # (same argument list: bindings/vvenc/enc_driver.cpp)
    L.vvref_encode_ex.restype = C.c_long
    L.vvref_encode_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_char_p, C.c_char_p, C.c_void_p, C.c_long, C.POINTER(C.c_double)]
    if hip:
        L.vvref_install_hip_hooks.argtypes = [C.c_int]
        L.vvref_hip_hook_calls.argtypes = [C.c_void_p]
        L.vvref_hip_hook_calls_ex.argtypes = [C.c_void_p, C.c_int]
    return L


def encode(L, yuv, width, height, in_bd, int_bd, preset=PRESET_FASTER, qp=32, threads=1, simd=None, options=None):
    """simd: None / "SCALAR" / "SSE41" / ... / "HIP[:mask]" (the reference's own switch); options: "name=value;..." for vvenc_set_param"""
    y, u, v = yuv
    out = np.zeros(8 << 20, np.uint8)
    secs = C.c_double()
    n = L.vvref_encode_ex(y.ctypes.data, u.ctypes.data, v.ctypes.data, width, height, y.shape[0], in_bd, int_bd, preset, qp, threads,
                          simd.encode() if simd else None, options.encode() if options else None, out.ctypes.data, out.size, C.byref(secs))
    assert n > 0, n
    bs = out[:n].tobytes()
    return hashlib.md5(bs).hexdigest(), n, secs.value
