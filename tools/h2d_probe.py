import torch, time
x = torch.rand(256,128,128)
pin = torch.empty(256,128,128).pin_memory()
dev = torch.empty(256,128,128, device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/n*1e3
print("host copy pageable->pinned  %.3f ms" % t(lambda: pin.copy_(x)))
print("host copy pageable->pageable %.3f ms" % t(lambda: x.clone()))
print("H2D pinned->dev             %.3f ms" % t(lambda: dev.copy_(pin, non_blocking=True)))
print("H2D pageable .to            %.3f ms" % t(lambda: x.to("cuda")))
print("threads", torch.get_num_threads())
import threading
res=[]
def bg():
    t0=time.perf_counter()
    for _ in range(20): pin.copy_(x)
    res.append((time.perf_counter()-t0)/20*1e3)
th=threading.Thread(target=bg); th.start(); th.join()
print("host copy in a thread        %.3f ms" % res[0])
