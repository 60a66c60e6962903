"""Copy rates of the box: 1 GiB host -> device and back from pinned and from pageable memory, and the host's own
pageable -> pinned memcpy (what bounds a staged upload).  Behind camcal_upload_bulk's design."""
import torch, time
dev = torch.device("cuda:0")
n = 1 << 30
hp = torch.empty(n, dtype=torch.uint8).pin_memory()
hq = torch.empty(n, dtype=torch.uint8)
d = torch.empty(n, dtype=torch.uint8, device=dev)
for name, h in (("pinned", hp), ("pageable", hq)):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("H2D %s 1 GiB: %.1f ms = %.1f GB/s" % (name, dt * 1e3, n / dt / 1e9), flush=True)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); h.copy_(d, non_blocking=True); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("D2H %s 1 GiB: %.1f ms = %.1f GB/s" % (name, dt * 1e3, n / dt / 1e9), flush=True)
t0 = time.perf_counter(); hp.copy_(hq); dt = time.perf_counter() - t0
print("host memcpy pageable -> pinned 1 GiB (torch, its threads): %.1f ms = %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
