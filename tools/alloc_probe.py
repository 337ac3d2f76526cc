"""Does device memory that other processes used before cost more to allocate?  Times raw hipMalloc (through the library's
ctx-free path: torch.cuda caching allocator bypassed with PYTORCH_NO_CUDA_MEMORY_CACHING) of 8 x 4 GiB and a fill of
each, (a) first on a fresh box, (b) after a child process has used and released 150 GiB."""
import os, subprocess, sys, time
os.environ["PYTORCH_NO_CUDA_MEMORY_CACHING"] = "1"
import torch


def alloc_round(tag):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bufs = [torch.empty(4 << 30, dtype=torch.uint8, device="cuda") for _ in range(8)]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for b in bufs:
        b.fill_(1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    del bufs
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"{tag}: alloc 32 GiB {t1 - t0:.3f}s, fill {t2 - t1:.3f}s, free {t3 - t2:.3f}s", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "child":
    big = [torch.ones(10 << 30, dtype=torch.uint8, device="cuda") for _ in range(15)]
    torch.cuda.synchronize()
    print("child used", sum(b.numel() for b in big) >> 30, "GiB", flush=True)
    sys.exit(0)
alloc_round("fresh box, round 1")
alloc_round("fresh box, round 2")
subprocess.run([sys.executable, __file__, "child"])
alloc_round("after a 150 GiB child, round 1")
alloc_round("after a 150 GiB child, round 2")
