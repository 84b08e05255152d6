import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recalgorithm_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def rel(a, b):
    b = b.double(); a = a.double()
    return float(((a - b).abs() / (b.abs() + b.pow(2).mean().sqrt())).max())
for (M, K, N) in [(4096, 416, 512), (4096, 512, 256), (4096, 256, 128)]:
    for use_mask in (False, True):
        for use_bn in (False, True):
            x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) / K ** 0.5
            g = torch.randn(M, N, device=dev); y = torch.randn(M, N, device=dev)
            dw = torch.zeros(K, N, device=dev); db = torch.zeros(N, device=dev)
            bn = None
            if use_bn:
                bx = torch.randn(M, K, device=dev); mean = bx.mean(0); rstd = 1.0 / (bx.var(0, unbiased=False) + 1e-3).sqrt()
                part = torch.zeros(ops.bn_partial_rows(M), 2 * K, device=dev)
                bn = (bx, mean, rstd, part)
            for rep in range(3):
                dx = ops.dense_bwd(x, g, y if use_mask else None, w, dw, db, defer=True, bn=bn)
                ops.flush_dense_splits()
                torch.cuda.synchronize()
                g2 = (g * (y > 0)) if use_mask else g
                e = (rel(dx, g2.double() @ w.double().t()), rel(dw, x.double().t() @ g2.double()), rel(db, g2.double().sum(0)))
                extra = ""
                if use_bn:
                    dxr = g2.double() @ w.double().t(); xh = (bx.double() - mean.double()) * rstd.double()
                    s = part.double().view(-1, 2, K).sum(0)
                    extra = f" bn sums {rel(s[0], dxr.sum(0)):.1e} {rel(s[1], (dxr * xh).sum(0)):.1e}"
                print(M, K, N, "mask", use_mask, "bn", use_bn, "rep", rep, "dx %.1e dw %.1e db %.1e" % e, extra, flush=True)
