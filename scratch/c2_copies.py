"""where do the copy / cat / elementwise kernels of the C2-shaped step come from?  python scratch/c2_copies.py [molecules]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hot_amd.batch.sparse import make_batch_concatenated
from hot_amd.models.encoder import Encoder
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = 'cuda'
rng = np.random.default_rng(1000)
nn_, ne_, eis = [], [], []
for _ in range(M):
    n = int(rng.integers(8, 21))
    und = {(i, i + 1) for i in range(n - 1)}
    while len(und) < int(round(1.05 * n)):
        i, j = sorted(int(v) for v in rng.integers(0, n, 2))
        if i != j:
            und.add((i, j))
    e = sorted([(i, j) for i, j in und] + [(j, i) for i, j in und])
    eis.append(np.array(e, dtype=np.int64).T); nn_.append(n); ne_.append(len(e))
ei = torch.from_numpy(np.concatenate(eis, axis=1)).to(dev)
g = torch.Generator(device=dev).manual_seed(77)
nf = torch.randn(sum(nn_), 256, device=dev, generator=g)
ef = torch.randn(sum(ne_), 256, device=dev, generator=g)
G = make_batch_concatenated(nf, ei, ef, nn_, ne_)
torch.manual_seed(0)
enc = Encoder(2, 0, [2] * 8, 512, 256, 256, 32, 32, 256, 4, 128, 128, 8, 'default', 'generalized_kernel', 0., 0., 0., sparse=True).to(dev)
enc.skip_redraw_projections = True
head = torch.nn.Linear(256, 1).to(dev)
params = list(enc.parameters()) + list(head.parameters())
def step():
    y = head(enc(G)).squeeze(-1)
    torch.nn.functional.l1_loss(y, torch.zeros_like(y)).backward()
    for p_ in params: p_.grad = None
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or ev.name.startswith('hotk') : continue
    if not ev.name.startswith('aten::'): continue
    if ev.cpu_children: 
        # only leaf aten ops (those that launch kernels themselves)
        if any(c.name.startswith('aten::') and c.device_time_total > 0 for c in ev.cpu_children): continue
    st = [s for s in (ev.stack or []) if 'hot_amd' in s or 'bench' in s or 'scratch' in s]
    key = (ev.name, str(ev.input_shapes)[:60], st[0][-70:] if st else '?')
    agg[key][0] += 1; agg[key][1] += ev.device_time_total
tot = sum(v[1] for v in agg.values())
print(f'aten leaf ops with device time: {tot/1e3:.2f} ms')
BYCOUNT = len(sys.argv) > 2
for k, v in sorted(agg.items(), key=lambda kv: (-kv[1][0] if BYCOUNT else -kv[1][1]))[:(90 if BYCOUNT else 45)]:
    print(f'{v[1]/1e3:8.3f} ms {v[0]:4d}x  {k[0]:28s} {k[1]:62s} {k[2]}')
