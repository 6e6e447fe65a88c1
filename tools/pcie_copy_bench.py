import time, torch
a = torch.rand(256, 16000).pin_memory()
w = torch.empty(256, 16200, device="cuda")
wh = torch.empty(256, 16200).pin_memory()
for _ in range(3):
    d = a.cuda(non_blocking=True); wh.copy_(w, non_blocking=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    d = a.cuda(non_blocking=True); wh.copy_(w, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print(f"H2D 16.4 MB + D2H 16.6 MB (pinned): {dt*1e3:.3f} ms per step  ({(a.numel()+w.numel())*4/dt/1e9:.1f} GB/s)")
