"""CPU model of csrc/conv3x3_wino16.hip's arithmetic (numpy): Winograd F(4,3) / F(2,3) ALONG Y on f16 x 2 pieces, with the kernel's
exact operation order (input transform B^T/4 in fp32 with the fmaf / add sequence of `item`, weight transform 4G in float64, the
three f16 x 2 products accumulated per 16-channel MFMA step, output transform A^T in fp32), against float64 conv2d, the direct
f16 x 2 kernel's arithmetic and a k-ordered fp32 fmaf chain (the exact-fp32-MFMA kernel).  Written BEFORE the kernel to decide
go / no-go on its numerics; `python scripts/sim_wino16_numerics.py [normal|relu|smooth|const]`."""
import sys

import numpy as np

kind = sys.argv[1] if len(sys.argv) > 1 else "normal"
rng = np.random.default_rng(0)
C, Co, H, W = 256, 32, 16, 32
x = rng.standard_normal((C, H + 2, W + 2)).astype(np.float32)
if kind == "relu":
    x = np.maximum(x, 0)
elif kind == "smooth":
    yy, xx = np.mgrid[0:H + 2, 0:W + 2]
    x = (np.abs(x) * 0.1 + 1.0 + 0.5 * np.sin(yy / 5.0 + np.arange(C)[:, None, None])).astype(np.float32)
elif kind == "const":
    x = np.full_like(x, 1.37)
x[:, 0, :] = 0; x[:, -1, :] = 0; x[:, :, 0] = 0; x[:, :, -1] = 0
w = (rng.standard_normal((Co, C, 3, 3)) / 48.0).astype(np.float32)
f32 = np.float32


def conv64():
    out = np.zeros((Co, H, W))
    for dy in range(3):
        for dx in range(3):
            out += np.einsum('oc,chw->ohw', w[:, :, dy, dx].astype(np.float64), x[:, dy:dy + H, dx:dx + W].astype(np.float64))
    return out


truth = conv64()


def fma(a, b, c):   # one rounding
    return (np.float64(a) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def split_x(v):
    x0 = v.astype(np.float16)
    x1 = ((v - x0.astype(np.float32)) * f32(2048.0)).astype(np.float16)
    return x0.astype(np.float64), x1.astype(np.float64)


def wscale(wm):
    _, e = np.frexp(f32(wm))
    return f32(2.0) ** (15 - e)


def split_w(wt, S):
    v = (wt * S).astype(np.float32)
    a = v.astype(np.float16)
    b = (v - a.astype(np.float32)).astype(np.float16)
    return a.astype(np.float64), b.astype(np.float64)


def mfma(acc, A, B):   # acc + sum of 16 products, one rounding (the matrix pipe's fused accumulate, to first order)
    return (acc.astype(np.float64) + np.einsum('ok,khw->ohw', A, B)).astype(np.float32)


def direct_fp32():
    acc = np.zeros((Co, H, W), np.float32)
    for c0 in range(0, C, 16):
        for dy in range(3):
            for dx in range(3):
                for k in range(c0, c0 + 16):
                    acc = (acc.astype(np.float64) + w[:, k, dy, dx].astype(np.float64)[:, None, None] * x[None, k, dy:dy + H, dx:dx + W].astype(np.float64)).astype(np.float32)
    return acc


def direct_f16x2():
    S = wscale(np.abs(w).max())
    wA, w1 = split_w(w, S)
    x0, x1 = split_x(x)
    acc = np.zeros((Co, H, W), np.float32)
    for c0 in range(0, C, 16):
        for dy in range(3):
            for dx in range(3):
                sl = (slice(c0, c0 + 16), slice(dy, dy + H), slice(dx, dx + W))
                acc = mfma(acc, w1[:, c0:c0 + 16, dy, dx], x0[sl])
                acc = mfma(acc, wA[:, c0:c0 + 16, dy, dx] / 2048.0, x1[sl])
                acc = mfma(acc, wA[:, c0:c0 + 16, dy, dx], x0[sl])
    return (acc * (f32(1.0) / S)).astype(np.float32)


def wino16(R):
    T, P = R + 2, H // R
    d = [x[:, i:i + R * P:R, :] for i in range(T)]   # d_i[c, p, col] = x[c, R p + i, col]
    g = w.astype(np.float64)
    g0, g1, g2 = g[:, :, 0], g[:, :, 1], g[:, :, 2]
    if R == 4:
        V = [None] * 6
        V[0] = fma(0.25, d[4], fma(-1.25, d[2], d[0]))
        Pq, Q = fma(0.25, d[4], -d[2]), fma(0.25, d[3], -d[1])
        V[1], V[2] = Pq + Q, Pq - Q
        Rr, Ss = (d[4] - d[2]) * f32(0.25), (d[3] - d[1]) * f32(0.5)
        V[3], V[4] = Rr + Ss, Rr - Ss
        V[5] = fma(0.25, d[5], fma(-1.25, d[3], d[1]))
        U = [g0, -(g0 + g1 + g2) * (2 / 3), -(g0 - g1 + g2) * (2 / 3), g0 / 6 + g1 / 3 + g2 * (2 / 3), g0 / 6 - g1 / 3 + g2 * (2 / 3), 4 * g2]
        bound = 4.0
    else:
        V = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]
        U = [g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2]
        bound = 2.0
    U = [u.astype(np.float32) for u in U]
    S = wscale(np.abs(w).max() * bound)
    assert max(np.abs(u).max() for u in U) * S < 65504
    m = []
    for t in range(T):
        uA, u1 = split_w(U[t], S)
        v0, v1 = split_x(V[t].astype(np.float32))
        acc = np.zeros((Co, P, W), np.float32)
        for c0 in range(0, C, 16):
            for dx in range(3):
                sl = (slice(c0, c0 + 16), slice(None), slice(dx, dx + W))
                acc = mfma(acc, u1[:, c0:c0 + 16, dx], v0[sl])
                acc = mfma(acc, uA[:, c0:c0 + 16, dx] / 2048.0, v1[sl])
                acc = mfma(acc, uA[:, c0:c0 + 16, dx], v0[sl])
        m.append(acc)
    out = np.zeros((Co, H, W), np.float32)
    if R == 4:
        s1, d1, s2, d2 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
        out[:, 0::4] = (m[0] + s1) + s2
        out[:, 1::4] = fma(2.0, d2, d1)
        out[:, 2::4] = fma(4.0, s2, s1)
        out[:, 3::4] = fma(8.0, d2, d1) + m[5]
    else:
        out[:, 0::2] = (m[0] + m[1]) + m[2]
        out[:, 1::2] = (m[1] - m[2]) - m[3]
    return (out * (f32(1.0) / S)).astype(np.float32)


def rep(name, got):
    e = got.astype(np.float64) - truth
    print("%-34s max %.3e rms %.3e" % (name, np.abs(e).max(), np.sqrt((e ** 2).mean())))


print("inputs:", kind, " output scale", float(np.abs(truth).max()))
rep("k-ordered fp32 fmaf chain (direct)", direct_fp32())
rep("direct f16 x 2 (split16)", direct_f16x2())
rep("Winograd F(2,3)-y on f16 x 2", wino16(2))
rep("Winograd F(4,3)-y on f16 x 2", wino16(4))
