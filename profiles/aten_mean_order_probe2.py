"""Search for ATen's mean(-1) summation order at large C (>= 8192): idx = x + B y + B Y g + B Y G it, vec-wide
accumulators, then (optionally) the G partial results combined, the Y tree, the X shuffles."""
import itertools

import torch

g = torch.Generator(device="cuda").manual_seed(0)


def tree(s, dim, mode):
    n = s.shape[dim]
    s = s.movedim(dim, -1)
    if mode == "inc":  # shuffle-down offsets 1, 2, 4, ...: lane 0 collects
        o = 1
        while o < n:
            s = s + torch.cat([s[..., o:], s[..., -o:]], dim=-1)
            o *= 2
        return s[..., 0]
    if mode == "dec":  # offsets n/2, ..., 1: element i += element i + off
        off = n // 2
        while off > 0:
            s = torch.cat([s[..., :off] + s[..., off:2 * off], s[..., off:]], dim=-1)
            off //= 2
        return s[..., 0]
    if mode == "seq":
        r = s[..., 0]
        for i in range(1, n):
            r = r + s[..., i]
        return r
    raise ValueError(mode)


def cand(p, B, Y, G, vec, ymode, xmode, gmode, order):
    rows, C = p.shape
    if C % (B * Y * G * vec):
        return None
    it = C // (B * Y * G * vec)
    v = p.view(rows, it, G, Y, B, vec)
    acc = v[:, 0].clone()
    for i in range(1, it):
        acc = acc + v[:, i]
    s = acc[..., 0]
    for j in range(1, vec):
        s = s + acc[..., j]  # [rows, G, Y, B]
    if order == "gyx":
        s = tree(s, 1, gmode) if G > 1 else s[:, 0]  # [rows, Y, B]
        s = tree(s, 1, ymode) if Y > 1 else s[:, 0]
        return tree(s, 1, xmode)
    if order == "yxg":
        s = tree(s, 2, ymode) if Y > 1 else s[:, :, 0]  # [rows, G, B]
        s = tree(s, 2, xmode)  # [rows, G]
        return tree(s, 1, gmode) if G > 1 else s[:, 0]
    if order == "ygx":  # per-cta y-reduce, then threads gather the G partials, then x
        s = tree(s, 2, ymode) if Y > 1 else s[:, :, 0]  # [rows, G, B]
        s = tree(s, 1, gmode) if G > 1 else s[:, 0]
        return tree(s, 1, xmode)


for rows, C in ((4096, 8192), (600, 14336), (2048, 16384)):
    p = torch.randn(rows, C, device="cuda", generator=g).pow(2)
    want = p.sum(-1)
    hits = []
    for B, Y, G, vec in itertools.product((32, 64, 128, 256, 512), (1, 2, 4, 8, 16), (1, 2, 4, 8), (1, 2, 4, 8)):
        if C % (B * Y * G * vec):
            continue
        for ymode, xmode, gmode, order in itertools.product(("dec", "inc"), ("inc", "dec"), ("seq", "dec", "inc"), ("gyx", "yxg", "ygx")):
            if G == 1 and (gmode != "seq" or order != "gyx"):
                continue
            if Y == 1 and ymode != "dec":
                continue
            s = cand(p, B, Y, G, vec, ymode, xmode, gmode, order)
            if s is not None and torch.equal(s, want):
                hits.append((B, Y, G, vec, ymode, xmode, gmode, order))
    print(rows, C, "hits:", hits[:8])
