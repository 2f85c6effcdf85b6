import sys, torch
sys.path.insert(0, '/root/repo')
from depthg_amd import ContrastiveCorrelationLoss
from oracle import depthg_oracle as O
dev = torch.device('cuda:0')
torch.manual_seed(0)
for dense, S, hw in ((True, 16, 16), (False, 14, 20)):
    B, C, D = 3, 1024, 70
    f, fp = torch.randn(B, C, hw, hw, device=dev), torch.randn(B, C, hw, hw, device=dev)
    c, cp = torch.randn(B, D, hw, hw, device=dev, requires_grad=True), torch.randn(B, D, hw, hw, device=dev, requires_grad=True)
    d = torch.randint(0, 256, (B, 1, 4 * hw, 4 * hw), device=dev).float()
    cfg = O.default_cfg(feature_samples=S, neg_samples=3, dim=D, dg_outputs="reduced", dg_dense_grid=dense)
    loss = ContrastiveCorrelationLoss(cfg)
    out = loss(f, fp, None, None, c, cp, d, d)
    loss.total.backward()
    print('dense' if dense else 'sampled', [round(float(o.mean()), 6) for o in out], float(c.grad.norm()), float(cp.grad.norm()))
