"""How fast do the library GEMMs (rocBLAS / hipBLASLt through torch) run the three skinny products
of the input-side aggregation path?  A yardstick for gemm_f32_kernel, not part of the product."""
import torch
dev = torch.device("cuda")
N, H, C, F = 2449029, 4, 64, 100
Z = torch.randn(N, H, F, device=dev)
g = torch.randn(N, H, C, device=dev)
W = torch.randn(H, C, F, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
Zt, gt = Z.transpose(0, 1), g.transpose(0, 1)          # [H, N, .] strided views
out = torch.empty(H, N, C, device=dev); q = torch.empty(H, N, F, device=dev); dw = torch.empty(H, C, F, device=dev)
print("out = Z W^T  (bmm, strided A):", t(lambda: torch.bmm(Zt, W.transpose(1, 2), out=out)))
print("q = g W      (bmm, strided A):", t(lambda: torch.bmm(gt, W, out=q)))
print("dW = g^T Z   (bmm, K = N)    :", t(lambda: torch.bmm(gt.transpose(1, 2), Zt, out=dw)))
Zc, gc = Zt.contiguous(), gt.contiguous()
print("contiguous per-head operands:", t(lambda: torch.bmm(Zc, W.transpose(1, 2), out=out)), t(lambda: torch.bmm(gc, W, out=q)),
      t(lambda: torch.bmm(gc.transpose(1, 2), Zc, out=dw)))
x = torch.randn(N, F, device=dev); Wf = torch.randn(H * C, F, device=dev)
print("h = x W^T [N,100]x[100,256]:", t(lambda: x @ Wf.t()))
