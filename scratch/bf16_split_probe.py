"""scratch: numerics of fp32 GEMMs emulated with split-bf16 MFMA products (fp32 accumulate), for the thin GEMMs whose
fp32 MFMA time is not hidden (entry GEMM [R,64]x[64,288]: 3.5 of 6.0 ms).  bf16 MFMA runs at 16x the fp32 MFMA rate, so
3 products (2-way split) cost 3/16 and 6 products (3-way split, terms down to 2^-24) 6/16 of the fp32 instruction time.
Prints max |err| / max |ref| against an fp64 product for fp32, 2-way and 3-way splits."""
import torch
torch.manual_seed(0)
dev = 'cuda'
R, K, C = 1 << 20, 64, 288
x = torch.randn(R, K, device=dev); w = torch.randn(K, C, device=dev) * K ** -0.5
ref = (x.double() @ w.double())
def split(t, n):
    parts, rest = [], t.clone()
    for _ in range(n):
        p = rest.bfloat16(); parts.append(p); rest = rest - p.float()
    return parts
def mm32(a, b):   # bf16 x bf16 -> fp32 accumulate and fp32 result
    try:
        return torch.mm(a, b, out_dtype=torch.float32)
    except TypeError:
        return a.float() @ b.float()      # same products, exact in fp32 (bf16 x bf16 fits 16 mantissa bits)
def emu(n, terms):
    xs, ws = split(x, n), split(w, n)
    out = torch.zeros(R, C, device=dev)
    for i, j in terms:
        out += mm32(xs[i], ws[j])
    return out
err = lambda y: float((y.double() - ref).abs().max() / ref.abs().max())
print('fp32 mm            ', err(x @ w))
print('bf16 1 product     ', err(emu(1, [(0, 0)])))
print('bf16 2-way, 3 prod ', err(emu(2, [(1, 0), (0, 1), (0, 0)])))
print('bf16 3-way, 6 prod ', err(emu(3, [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)])))
