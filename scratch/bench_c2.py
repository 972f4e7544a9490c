"""scratch: C2-shaped step: enc-small Encoder on PCQM-like synthetic molecules, fwd + L1 loss backward."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hot_amd.batch.sparse import make_batch_concatenated
from hot_amd.models.encoder import Encoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = 'cuda'
rng = np.random.default_rng(0)
nn_, ne_, eis = [], [], []
for _ in range(B):
    n = int(rng.integers(8, 21))
    m = int(round(1.05 * n))
    und = set()
    while len(und) < m:
        i, j = rng.integers(0, n, 2)
        if i != j:
            und.add((min(i, j), max(i, j)))
    e = sorted([(i, j) for i, j in und] + [(j, i) for i, j in und])
    eis.append(np.array(e, dtype=np.int64).T); nn_.append(n); ne_.append(len(e))
ei = torch.from_numpy(np.concatenate(eis, axis=1)).to(dev)
nf = torch.randn(sum(nn_), 256, device=dev); ef = torch.randn(sum(ne_), 256, device=dev)
G = make_batch_concatenated(nf, ei, ef, nn_, ne_)
torch.manual_seed(0)
if os.environ.get('HOTK_C2_ARCH', 'enc-small') == 'enc':      # enc.sh: dim_hidden 512, dim_qk = dim_v = 64 (d = 16, m = 44)
    enc = Encoder(2, 0, [2] * 8, 512, 256, 512, 64, 64, 512, 4, 256, 256, 16, 'default', 'generalized_kernel', 0., 0., 0., sparse=True).to(dev)
else:
    enc = Encoder(2, 0, [2] * 8, 512, 256, 256, 32, 32, 256, 4, 128, 128, 8, 'default', 'generalized_kernel', 0., 0., 0., sparse=True).to(dev)
enc.skip_redraw_projections = True
STORAGE = os.environ.get('HOTK_C2_STORAGE', 'f32')      # bf16: the fused 2->2 layers in their bf16 storage mode
for lyr in enc.layers:
    if hasattr(lyr.attn, 'set_storage'):
        lyr.attn.set_storage(STORAGE)
head = torch.nn.Linear(256, 1).to(dev)
params = list(enc.parameters()) + list(head.parameters())
print('storage', STORAGE, 'graphs', B, 'entries', sum(G.n_edges), 'padded', G.values.shape[0] * G.values.shape[1], 'params %.2f M' % (sum(p.numel() for p in params) / 1e6))
def step():
    y = head(enc(G)).squeeze(-1)
    loss = torch.nn.functional.l1_loss(y, torch.zeros_like(y))
    loss.backward()
    for p in params: p.grad = None
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 10
for _ in range(K): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print('step %.2f ms  -> %.3e entries/s' % (dt * 1e3, sum(G.n_edges) / dt))

# the same step captured once into a HIP graph (torch.cuda.CUDAGraph) and replayed: at C2's size (a few thousand entries
# per GPU, ~9 layers) the eager step is bound by ~10^3 kernel launches, not by the kernels
if os.environ.get('HOTK_C2_GRAPH', '1') == '1':
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = head(enc(G)).squeeze(-1)
        loss = torch.nn.functional.l1_loss(y, torch.zeros_like(y))
        loss.backward()
    eager_loss = float(loss)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): g.replay()
    torch.cuda.synchronize(); dtg = (time.perf_counter() - t0) / K
    gn = sum(float(p.grad.abs().sum()) for p in params if p.grad is not None)
    print('graph replay %.2f ms -> %.3e entries/s (loss %.6f, grad l1 %.4e)' % (dtg * 1e3, sum(G.n_edges) / dtg, float(loss), gn))
