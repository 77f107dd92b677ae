"""Diagnostic: what this GPU sustains for plain streaming writes / copies (torch fill_ / copy_), to put k_step's
write-heavy traffic (profiles/r1_hbm_traffic.json) into perspective."""
import torch, time
dev = "cuda"
for mb in (64, 256, 1024, 4096):
    n = mb << 20
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
    for name, fn, bytes_ in (("fill (write only)", lambda: a.fill_(1), n), ("copy (read + write)", lambda: b.copy_(a), 2 * n),
                             ("sum (read only)", lambda: a.view(torch.int32).sum(), n)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        reps = 20
        for _ in range(reps): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
        print("%5d MiB %-20s %7.1f us  %6.2f TB/s" % (mb, name, dt * 1e6, bytes_ / dt / 1e12))
