"""scratch: streaming ceilings on this GPU: fill (write only), copy (read + write), sum (read only)"""
import torch, time
dev = 'cuda'
n = 3_200_000_000  # floats = 12.8 GB
x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
tf = t(lambda: x.fill_(1.0)); print('fill  %.2f ms  %.2f TB/s written' % (tf * 1e3, 4 * n / tf / 1e12))
tc = t(lambda: y.copy_(x)); print('copy  %.2f ms  %.2f TB/s total (r+w)' % (tc * 1e3, 8 * n / tc / 1e12))
ts = t(lambda: x.sum()); print('sum   %.2f ms  %.2f TB/s read' % (ts * 1e3, 4 * n / ts / 1e12))
