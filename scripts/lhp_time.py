"""Time the LHP propagations at the headline map (B=32, D=70, 28x28, 6 heads): python scripts/lhp_time.py [B]"""
import sys
import torch
sys.path.insert(0, ".")
from depthg_amd import ops  # noqa: E402
from depthg_amd.lhp import neighbour_counts  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
D, sz, heads = 70, 28, 6
P = sz * sz
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
code = torch.randn(B, D, sz, sz, device=dev, generator=g)
attn = torch.softmax(2 * torch.randn(B, heads, P + 1, P + 1, device=dev, generator=g), -1)
depth = torch.rand(B, 1, 224, 224, device=dev, generator=g) * 200
up = torch.randn(B, D, sz, sz, device=dev, generator=g)
div = neighbour_counts(sz).float().to(dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out, wmap = ops.lhp_map_forward(ops.LHP_ATTN, code, attn=attn)
print("attn bytes %.0f MB" % (attn.numel() * 4 / 1e6))
print("ATTN fwd %.1f us" % timed(lambda: ops.lhp_map_forward(ops.LHP_ATTN, code, attn=attn)))
print("ATTN bwd %.1f us" % timed(lambda: ops.lhp_map_backward(ops.LHP_ATTN, up, wmap)))
o2, w9 = ops.lhp_map_forward(ops.LHP_ORIG_ATTN, code, attn=attn, divide=div)
print("ORIG_ATTN fwd %.1f us" % timed(lambda: ops.lhp_map_forward(ops.LHP_ORIG_ATTN, code, attn=attn, divide=div)))
print("ORIG_ATTN bwd %.1f us" % timed(lambda: ops.lhp_map_backward(ops.LHP_ORIG_ATTN, up, w9, div)))
print("ORIG_DEPTH fwd %.1f us" % timed(lambda: ops.lhp_map_forward(ops.LHP_ORIG_DEPTH, code, depth=depth, divide=div)))
o3, pts, st = ops.lhp_forward(code, depth)
print("depth strategy fwd %.1f us" % timed(lambda: ops.lhp_forward(code, depth)))
print("depth strategy bwd %.1f us" % timed(lambda: ops.lhp_backward(up, pts, st)))
# dense check of the full-size map against torch on the GPU
want = torch.einsum("bpq,bdq->bdp", wmap, code.reshape(B, D, P)).reshape(B, D, sz, sz) / P
print("fwd vs map einsum rel %.2e" % float((out - want).norm() / want.norm()))
gb = ops.lhp_map_backward(ops.LHP_ATTN, up, wmap)
gw = torch.einsum("bpq,bdp->bdq", wmap, up.reshape(B, D, P)).reshape(B, D, sz, sz) / P
print("bwd vs map einsum rel %.2e" % float((gb - gw).norm() / gw.norm()))
print("zeros per row min/max", int((wmap == 0).sum(-1).min()), int((wmap == 0).sum(-1).max()))
