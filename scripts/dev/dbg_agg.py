"""Developer aid: planned aggregation against the plain fixed-k kernel, row by row (which rows / columns differ).
    python scripts/dev/dbg_agg.py"""
import torch, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from cosmology_gnn_simulation_amd import ops
DEV='cuda'
def _knn_senders(n, k, seed):
    gen = torch.Generator().manual_seed(seed)
    pos = torch.rand(n, 3, generator=gen).to(DEV)
    snd, _, order = ops.knn_periodic(pos, 1.0, k, want_edge_attr=False, want_order=True)
    inv = torch.empty_like(order)
    inv[order.long()] = torch.arange(n, device=DEV, dtype=order.dtype)
    return inv[snd.view(n, k)[order.long()].long()].reshape(-1).contiguous().int()
for n,k,w in [(20000,16,128),(9001,16,128),(16384,16,256)]:
    src=_knn_senders(n,k,n+k)
    x=torch.randn(n,w,device=DEV)
    plain=ops.aggregate(x,src,None,n,k)
    plan=ops.AggregatePlan(src,n,k)
    got=ops.aggregate(x,src,None,n,k,plan=plan)
    bad=(got!=plain).any(dim=1).nonzero().flatten()
    print(n,k,w,'bad rows',bad.numel(), bad[:10].tolist(), bad[-5:].tolist())
    if bad.numel():
        r=int(bad[0]); cols=(got[r]!=plain[r]).nonzero().flatten()
        print(' row',r,'cols',cols[:8].tolist(), got[r,cols[:4]].tolist(), plain[r,cols[:4]].tolist())
