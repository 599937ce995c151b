"""Which summation order does ATen's mean(-1) use for a contiguous fp32 [rows, C] tensor on this build?  Candidates are
emulated with torch ops (explicit fp32 adds in a fixed order) and compared bitwise.  (GPU box)"""
import itertools

import torch

g = torch.Generator(device="cuda").manual_seed(0)


def cand(p, B, vec, tree, fold):
    rows, C = p.shape
    if C % (B * vec):
        return None
    it = C // (B * vec)
    v = p.view(rows, it, B, vec)
    acc = v[:, 0].clone()
    for i in range(1, it):
        acc = acc + v[:, i]
    if fold == "seq":
        s = acc[..., 0]
        for j in range(1, vec):
            s = s + acc[..., j]
    else:  # pairwise
        a = acc
        while a.shape[-1] > 1:
            a = a[..., 0::2] + a[..., 1::2]
        s = a[..., 0]
    # lanes: s [rows, B]
    offs = [1 << k for k in range(B.bit_length() - 1)]
    if tree == "dec":
        offs = offs[::-1]
    for o in offs:
        sh = torch.cat([s[:, o:], s[:, -o:]], dim=1)  # shfl_down: lane i reads lane i + o (tail values irrelevant for lane 0)
        s = s + sh
    return s[:, 0]


def cand_split(p, Y=8, B=64, vec=4):
    """thread (x, y) takes elements (x + B y + B Y it) vec + j; y-tree with offsets Y/2 .. 1; then the x shuffles (increasing)."""
    rows, C = p.shape
    if C % (B * Y * vec):
        return None
    it = C // (B * Y * vec)
    v = p.view(rows, it, Y, B, vec)
    acc = v[:, 0].clone()
    for i in range(1, it):
        acc = acc + v[:, i]
    s = acc[..., 0]
    for j in range(1, vec):
        s = s + acc[..., j]          # [rows, Y, B]
    off = Y // 2
    while off > 0:
        s = torch.cat([s[:, :off] + s[:, off:2 * off], s[:, off:]], dim=1)
        off //= 2
    s = s[:, 0]
    for o in (1, 2, 4, 8, 16, 32):
        s = s + torch.cat([s[:, o:], s[:, -o:]], dim=1)
    return s[:, 0]


for rows, C in ((2048, 4096), (8192, 4096), (2048, 2048), (2048, 5120), (4096, 8192), (600, 14336), (231, 512), (2048, 3584), (2048, 7168), (2048, 6144), (2048, 16384)):
    x = torch.randn(rows, C, device="cuda", generator=g)
    p = x.pow(2)
    want_sum = p.sum(-1)
    want_mean = p.mean(-1)
    hits = []
    for B, vec, tree, fold in itertools.product((16, 32, 64, 128, 256, 512, 1024), (1, 2, 4, 8), ("inc", "dec"), ("seq", "pair")):
        s = cand(p, B, vec, tree, fold)
        if s is None:
            continue
        if torch.equal(s, want_sum):
            hits.append((B, vec, tree, fold))
    for Y in (2, 4, 8, 16):
        sp = cand_split(p, Y)
        if sp is not None and torch.equal(sp, want_sum):
            hits.append(("split", Y))
    fac = torch.tensor(float(rows) / float(rows * C), device="cuda")
    print(rows, C, "sum order hits:", hits[:6], "| mean == sum * fl(1/C):", torch.equal(want_sum * fac, want_mean),
          "| mean == sum / C:", torch.equal(want_sum / C, want_mean))
