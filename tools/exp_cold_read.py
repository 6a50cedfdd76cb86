import torch, time
dev='cuda:0'
ws=[torch.randn(24576*768//2, device=dev) for _ in range(24)]  # 37.7 MB each (fp32 words = bf16 pairs)
small=[torch.randn(768*512//2, device=dev) for _ in range(24)]
for name, arrs in (("38MB cold (24 buffers cycling)", ws), ("0.8MB", small)):
    for _ in range(2):
        for w in arrs: w.sum()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        for w in arrs: w.sum()
    e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1)/120*1e3, "us per sum")
one=ws[0]
for _ in range(3): one.sum()
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): one.sum()
e1.record(); torch.cuda.synchronize()
print("38MB hot (same buffer)", e0.elapsed_time(e1)/100*1e3, "us per sum")
