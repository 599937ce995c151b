"""Independent decoder of llama.cpp's block_q{2,3,4,5,6}_K layouts (ggml-quants.c
dequantize_row_q*_K), written from the format spec, NOT from the reference packers.

unpack(q_type, bytes[R, nb*type_size]) -> (codes[R,C] int, d, sc[R,C/G] int, dmin, mn[R,C/G] int)
where the dequantized value is  d*sc*code - dmin*mn  (Q3_K/Q6_K: code and sc signed,
dmin = mn = 0).  Used for pack -> unpack round trips at any size.
"""
import numpy as np

TS = {10: 84, 11: 110, 12: 144, 13: 176, 14: 210}


def _f16(b):  # [..., 2] uint8 -> uint16 bits
    return b[..., 0].astype(np.uint16) | (b[..., 1].astype(np.uint16) << 8)


def _scale_min_k4(sc12):
    """12 packed bytes -> (8 scales, 8 mins), ggml get_scale_min_k4."""
    q = sc12.astype(np.int32)
    sc = np.empty(q.shape[:-1] + (8,), np.int32)
    mn = np.empty_like(sc)
    for j in range(8):
        if j < 4:
            sc[..., j] = q[..., j] & 63
            mn[..., j] = q[..., j + 4] & 63
        else:
            sc[..., j] = (q[..., j + 4] & 0xF) | ((q[..., j - 4] >> 6) << 4)
            mn[..., j] = (q[..., j + 4] >> 4) | ((q[..., j] >> 6) << 4)
    return sc, mn


def unpack(q_type, packed):
    R = packed.shape[0]
    ts = TS[q_type]
    nb = packed.shape[1] // ts
    b = packed.reshape(R * nb, ts).astype(np.int32)
    n = b.shape[0]
    codes = np.zeros((n, 256), np.int32)
    zeros16 = np.zeros(n, np.uint16)
    if q_type == 10:  # scales[16] qs[64] d dmin
        scales, qs = b[:, :16], b[:, 16:80]
        d, dmin = _f16(b[:, 80:82]), _f16(b[:, 82:84])
        sc, mn = scales & 0xF, scales >> 4
        for ch in range(2):
            for k in range(4):
                codes[:, ch * 128 + 32 * k: ch * 128 + 32 * k + 32] = (qs[:, ch * 32: ch * 32 + 32] >> (2 * k)) & 3
    elif q_type == 11:  # hmask[32] qs[64] scales[12] d
        hm, qs, sb = b[:, :32], b[:, 32:96], b[:, 96:108]
        d, dmin = _f16(b[:, 108:110]), zeros16
        sc = np.zeros((n, 16), np.int32)
        for j in range(16):
            lo = (sb[:, j] & 0xF) if j < 8 else (sb[:, j - 8] >> 4)
            hi = (sb[:, 8 + j % 4] >> (2 * (j // 4))) & 3
            sc[:, j] = (lo | (hi << 4)) - 32
        mn = np.zeros_like(sc)
        for ch in range(2):
            for k in range(4):
                low2 = (qs[:, ch * 32: ch * 32 + 32] >> (2 * k)) & 3
                bit = (hm >> (ch * 4 + k)) & 1
                codes[:, ch * 128 + 32 * k: ch * 128 + 32 * k + 32] = low2 - np.where(bit == 1, 0, 4)
    elif q_type in (12, 13):  # d dmin scales[12] [qh[32]] qs[128]
        d, dmin = _f16(b[:, 0:2]), _f16(b[:, 2:4])
        sc, mn = _scale_min_k4(b[:, 4:16])
        if q_type == 12:
            qs = b[:, 16:144]
            qh = None
        else:
            qh, qs = b[:, 16:48], b[:, 48:176]
        for j in range(4):
            lo = qs[:, 32 * j: 32 * j + 32] & 0xF
            hi = qs[:, 32 * j: 32 * j + 32] >> 4
            if qh is not None:
                lo = lo + np.where((qh >> (2 * j)) & 1, 16, 0)
                hi = hi + np.where((qh >> (2 * j + 1)) & 1, 16, 0)
            codes[:, 64 * j: 64 * j + 32] = lo
            codes[:, 64 * j + 32: 64 * j + 64] = hi
    elif q_type == 14:  # ql[128] qh[64] scales[16] d
        ql, qh = b[:, :128], b[:, 128:192]
        sc = b[:, 192:208].astype(np.uint8).view(np.int8).astype(np.int32).reshape(n, 16)
        mn = np.zeros_like(sc)
        d, dmin = _f16(b[:, 208:210]), zeros16
        for ch in range(2):
            l_ = ql[:, 64 * ch: 64 * ch + 64]
            h_ = qh[:, 32 * ch: 32 * ch + 32]
            base = 128 * ch
            codes[:, base: base + 32] = ((l_[:, :32] & 0xF) | (((h_ >> 0) & 3) << 4)) - 32
            codes[:, base + 32: base + 64] = ((l_[:, 32:] & 0xF) | (((h_ >> 2) & 3) << 4)) - 32
            codes[:, base + 64: base + 96] = ((l_[:, :32] >> 4) | (((h_ >> 4) & 3) << 4)) - 32
            codes[:, base + 96: base + 128] = ((l_[:, 32:] >> 4) | (((h_ >> 6) & 3) << 4)) - 32
    else:
        raise ValueError(q_type)
    G = 256 // sc.shape[1]
    return (codes.reshape(R, nb * 256), d.reshape(R, nb), sc.reshape(R, nb * (256 // G)), dmin.reshape(R, nb),
            mn.reshape(R, nb * (256 // G)))


# ---- Q8_0 (ggml-quants.c quantize_row_q8_0_ref / dequantize_row_q8_0), scalar, from the C source's definition ----
def q8_0_encode_scalar(x):
    """x: float32 [n], n % 32 == 0 -> bytes.  One Python loop per element (small inputs only): fp32 arithmetic through
    np.float32 scalars, roundf as floor(|v| + 0.5) in exact double arithmetic with the sign restored."""
    import math
    import struct
    x = np.asarray(x, np.float32).ravel()
    out = bytearray()
    for b in range(0, x.size, 32):
        blk = x[b:b + 32]
        amax = np.float32(0)
        for v in blk:
            amax = max(amax, np.float32(abs(v)))
        d = np.float32(amax / np.float32(127))
        inv = np.float32(1) / d if d != 0 else np.float32(0)
        out += np.float16(d).tobytes()
        for v in blk:
            x0 = float(np.float32(v * inv))
            out += struct.pack("<b", int(math.copysign(math.floor(abs(x0) + 0.5), x0)))
    return bytes(out)


def q8_0_decode(raw, n):
    """bytes -> float32 [n]: y = d * q."""
    b = np.frombuffer(raw, np.uint8).reshape(-1, 34)
    d = b[:, :2].copy().view(np.float16).astype(np.float32)
    q = b[:, 2:].copy().view(np.int8).astype(np.float32)
    return (d * q).reshape(-1)[:n]
