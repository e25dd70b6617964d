"""Where the time of a fed step goes: host-side duration of every phase of the feeder loop (debug aid)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ava_amd import synthetic as syn
from ava_amd.vae import VAE
B = 256
pool = [torch.from_numpy(syn.spectrograms(B, start_item=i * B)) for i in range(4)]
model = VAE(z_dim=32, device_name="cuda")
model.train()
def step(d):
    model.optimizer.zero_grad(); model._forward_device(d, need_grad=True, accumulate=True); model._backward_device(d); model.optimizer.step()
for _ in range(5): step(pool[0].cuda())
torch.cuda.synchronize()
T = collections.defaultdict(float)
def tm(name, fn):
    t0 = time.perf_counter(); r = fn(); T[name] += time.perf_counter() - t0; return r
cs = torch.cuda.Stream()
pins = [torch.empty_like(pool[0]).pin_memory() for _ in range(3)]
devs = [torch.empty_like(pool[0], device="cuda") for _ in range(3)]
ready = [torch.cuda.Event() for _ in range(3)]; rel = [torch.cuda.Event() for _ in range(3)]
mode = sys.argv[1] if len(sys.argv) > 1 else "stream"
N = 30
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(N):
    s = k % 3
    if mode == "side":
        tm("rel.sync", rel[s].synchronize)
        def h2d():
            with torch.cuda.stream(cs):
                devs[s].copy_(pool[k % 4], non_blocking=True); ready[s].record(cs)
        tm("h2d pageable", h2d)
        tm("wait_event", lambda: torch.cuda.current_stream().wait_event(ready[s]))
        tm("step launches", lambda: step(devs[s]))
        tm("rel.record", lambda: rel[s].record(torch.cuda.current_stream()))
        continue
    if mode == "sync":
        d = tm("to(device)", lambda: pool[k % 4].to("cuda"))
        tm("step launches", lambda: step(d))
        continue
    tm("ready.sync", ready[s].synchronize)
    tm("host copy", lambda: pins[s].copy_(pool[k % 4]))
    if mode == "stream":
        def h2d():
            with torch.cuda.stream(cs):
                cs.wait_event(rel[s]); devs[s].copy_(pins[s], non_blocking=True); ready[s].record(cs)
        tm("h2d enqueue", h2d)
        tm("wait_event", lambda: torch.cuda.current_stream().wait_event(ready[s]))
    else:
        tm("h2d enqueue", lambda: devs[s].copy_(pins[s], non_blocking=True))
    tm("step launches", lambda: step(devs[s]))
    tm("rel.record", lambda: rel[s].record(torch.cuda.current_stream()))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("mode %s: %.3f ms/step" % (mode, dt / N * 1e3))
for k_, v in T.items(): print("  %-14s %.3f ms/step" % (k_, v / N * 1e3))
