import torch, time
x = torch.empty(int(10.5e9 // 4), dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for _ in range(2): y.copy_(x)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): y.copy_(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f"D2D copy of {x.numel()*4/1e9:.1f} GB: {dt*1e3:.2f} ms = {2*x.numel()*4/dt/1e12:.2f} TB/s (read+write)")
z = torch.empty(int(21e9 // 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): z.zero_()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"fill 21 GB: {dt*1e3:.2f} ms = {z.numel()*4/dt/1e12:.2f} TB/s")
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): s = z.sum()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"read 21 GB (sum): {dt*1e3:.2f} ms = {z.numel()*4/dt/1e12:.2f} TB/s")
