import torch, time
x = torch.empty(135*(1<<20), dtype=torch.int64).pin_memory()
d = torch.empty_like(x, device="cuda")
for chunk in (len(x), 16<<20, 4<<20):
    torch.cuda.synchronize(); t=time.perf_counter()
    for i in range(0, len(x), chunk):
        d[i:i+chunk].copy_(x[i:i+chunk], non_blocking=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print("chunk %d MiB: %.1f ms, %.1f GB/s" % (chunk*8>>20, dt*1e3, x.numel()*8/dt/1e9))
