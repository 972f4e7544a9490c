import torch, sys
R = 1000000; dev = 'cuda'
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def splitk(A, B, S):
    R_, K = A.shape; N = B.shape[1]
    rs = (R_ // S)
    main = rs * S
    out = torch.bmm(A[:main].view(S, rs, K).transpose(1, 2), B[:main].view(S, rs, N)).sum(0)
    if main < R_:
        out = out + A[main:].t() @ B[main:]
    return out
for (K, N) in [(32, 320), (320, 32), (160, 288)]:
    A = torch.randn(R, K, device=dev); B = torch.randn(R, N, device=dev)
    ref = A.t() @ B
    print(f'K={K} N={N} plain {timeit(lambda: A.t() @ B):.3f} ms')
    for S in (64, 256, 1024, 4096):
        o = splitk(A, B, S)
        err = ((o - ref).abs().max() / ref.abs().max()).item()
        print(f'   S={S:5d} {timeit(lambda: splitk(A, B, S)):.3f} ms  err {err:.1e}')
