"""What a plain elementwise kernel reaches on this device with the byte mix of the streaming pointwise kernels (development aid)."""
import torch
dev = torch.device('cuda:0')
def timed(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
a = torch.randn(42 * 1024 * 1024, device=dev)      # 168 MB
b = torch.randn_like(a); c = torch.empty_like(a)
t = timed(lambda: torch.add(a, b, out=c)); print(f'c = a + b (2 reads + 1 write of 168 MB): {t:.1f} us  {3 * a.numel() * 4 / t / 1e6:.2f} TB/s')
t = timed(lambda: c.copy_(a)); print(f'c = a (1 read + 1 write): {t:.1f} us  {2 * a.numel() * 4 / t / 1e6:.2f} TB/s')
t = timed(lambda: a.sum()); print(f'sum(a) (1 read): {t:.1f} us  {a.numel() * 4 / t / 1e6:.2f} TB/s')
t = timed(lambda: c.zero_()); print(f'c = 0 (1 write): {t:.1f} us  {a.numel() * 4 / t / 1e6:.2f} TB/s')
