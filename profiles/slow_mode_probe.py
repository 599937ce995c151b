"""Minimal reproducer attempt for the "slow mode" of DESIGN.md 6b: the main thread runs a chain of small kernels that hop
between two streams through events (what the Cholesky leaves / column-loop kernels do); a second thread launches one tiny
kernel per 50 ms on (a) nothing, (b) the null stream, (c) a stream of its own.  Prints the main loop's time per hop."""
import threading
import time

import torch

dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def hops(n):
    ev = torch.cuda.Event()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        a, b = (s1, s2) if i & 1 else (s2, s1)
        with torch.cuda.stream(a):
            a.wait_event(ev)
            x.add_(1.0)
            ev = torch.cuda.Event()
            ev.record(a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def side(mode, stop):
    st = torch.cuda.Stream(dev) if mode == "own" else None
    y = torch.zeros(16, device=dev)
    while not stop.is_set():
        if mode == "own":
            with torch.cuda.stream(st):
                y.add_(1.0)
        elif mode == "null":
            y.add_(1.0)
        time.sleep(0.05)


print("warm", round(hops(2000), 2), "us per hop")
for mode in ("none", "null", "own", "none"):
    stop = threading.Event()
    th = threading.Thread(target=side, args=(mode, stop))
    th.start()
    time.sleep(0.3)
    r = [round(hops(4000), 2) for _ in range(3)]
    stop.set()
    th.join()
    print(f"side thread: {mode:5s} -> {r} us per cross-stream hop")
