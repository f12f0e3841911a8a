"""HBM bandwidth probe with torch kernels: fill (write only), sum (read only), copy (read + write); sizes far above the 256 MB Infinity Cache."""
import torch, time
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.empty_like(a)
    a.fill_(1.0)
    tw = t(lambda: a.fill_(2.0)); tr = t(lambda: a.view(torch.int16).sum()); tc = t(lambda: b.copy_(a))
    gb = mb / 1024.0
    print(f"{mb:5d} MB  write {gb/tw/1e3:6.2f} TB/s   read (torch sum) {gb/tr/1e3:6.2f} TB/s   copy (r+w) {2*gb/tc/1e3:6.2f} TB/s", flush=True)
