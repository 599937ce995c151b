"""How fast does a device tensor reach the writer's staging slot?  shared-memory slot + cudaHostRegister (the
_Saver's way) vs an unregistered shared slot vs torch pinned memory.  usage (GPU box): python profiles/save_copy_probe.py"""
import time

import torch

n = 600 << 20
src = torch.empty(n, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()


def rate(dst, label):
    for _ in range(2):
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            dst.copy_(src, non_blocking=True)
            st.synchronize()
        dt = time.perf_counter() - t0
    print(f"{label:34s} {n / dt / 1e9:6.2f} GB/s ({dt * 1e3:.1f} ms)")


shm = torch.empty(n, dtype=torch.uint8).share_memory_()
rate(shm, "shared memory, not registered")
rc = torch.cuda.cudart().cudaHostRegister(shm.data_ptr(), shm.numel(), 0)
print("cudaHostRegister ->", rc, int(rc))
rate(shm, "shared memory, registered")
rate(torch.empty(n, dtype=torch.uint8, pin_memory=True), "torch pinned")
rate(torch.empty(n, dtype=torch.uint8), "pageable")
