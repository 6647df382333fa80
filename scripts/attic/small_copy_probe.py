"""after another GPU process: how long do a small D2H copy to pageable / pinned memory and an idle synchronize take? (torch only)"""
import time, torch
d = torch.arange(4096, device="cuda", dtype=torch.int64)
pin = torch.empty(4096, dtype=torch.int64).pin_memory()
big = torch.empty(1 << 28, device="cuda", dtype=torch.uint8)
torch.cuda.synchronize()
def t(f, n=5):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); a = time.perf_counter(); f(); out.append((time.perf_counter() - a) * 1e3)
    return " ".join("%.2f" % x for x in out)
print("idle synchronize      :", t(torch.cuda.synchronize))
print("D2H 32 KB -> pageable :", t(lambda: d.cpu()))
print("D2H 32 KB -> pinned   :", t(lambda: (pin.copy_(d, non_blocking=True), torch.cuda.synchronize())))
print("kernel + synchronize  :", t(lambda: (big.fill_(1), torch.cuda.synchronize())))
print("kernel, D2H pageable  :", t(lambda: (big.fill_(1), d.cpu())))
print("kernel, D2H pinned    :", t(lambda: (big.fill_(1), pin.copy_(d, non_blocking=True), torch.cuda.synchronize())))
