import torch, sys, time
R = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
dev = 'cuda'
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = {  # name: (M, K, N) for out[M,N] = A[M,K] @ B[K,N]
 'fc2_fwd  X5[R,160]@W[160,288]': (R, 160, 288),
 'fc2_dX   dh2[R,288]@Wt[288,160]': (R, 288, 160),
 'fc2_dW   X5t[160,R]@dh2[R,288]': (160, R, 288),
 'fcv_fwd  x[R,32]@Wv[32,320]': (R, 32, 320),
 'fcv_dx   dv[R,320]@Wv[320,32]': (R, 320, 32),
 'fcv_dW   dvt[320,R]@x[R,32]': (320, R, 32),
 'fco_fwd  att[R,320]@Wo[320,32]': (R, 320, 32),
 'fco_datt dy[R,32]@Wo[32,320]': (R, 32, 320),
 'fco_dW   dyt[32,R]@att[R,320]': (32, R, 320),
}
for lib in ('default', 'hipblaslt', 'hipblas'):
    try:
        if lib != 'default': torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, 'unavailable', e); continue
    tot = 0
    for name, (M, K, N) in shapes.items():
        if M == R:
            A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev)
            f = lambda: A @ B
        else:
            At = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
            f = lambda: At.t() @ B
        ms = timeit(f); tot += ms
        print(f'{lib:10s} {name:34s} {ms:8.3f} ms  {2*M*K*N/ms/1e9:8.1f} TFLOP/s')
    print(lib, 'total', round(tot, 3))
