"""How fast is the vendor fp32 GEMM on the encode shape?  (A ceiling reference, not part of the product.)"""
import torch
dev = "cuda"
torch.backends.cuda.matmul.allow_tf32 = False
for M, K, N in ((16384, 768, 24576), (65536, 768, 24576), (8192, 8192, 8192)):
    a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(5): torch.matmul(a, b, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.matmul(a, b, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"torch.matmul fp32 {M}x{K}x{N}: {ms:.3f} ms  {2.0*M*K*N/ms/1e9:.1f} TFLOP/s")
