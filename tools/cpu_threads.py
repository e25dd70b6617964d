import sys, time, torch
sys.path.insert(0, '.')
from ava_amd import synthetic as syn
from oracle import vae_oracle as O
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    P = O.to_params(syn.fixture_parameters(32), requires_grad=True)
    running = O.fresh_running_stats(); opt = {"step": 0, "m": {}, "v": {}}
    x = torch.from_numpy(syn.spectrograms(256)); ew, ed = [torch.from_numpy(a) for a in syn.noise(256, 32)]
    O.train_step(P, x, ew, ed, running, opt)
    t0 = time.perf_counter()
    for _ in range(2): O.train_step(P, x, ew, ed, running, opt)
    dt = (time.perf_counter() - t0) / 2
    print(th, 'threads: %.0f ms/step, %.1f spectrograms/s' % (dt * 1e3, 256 / dt), flush=True)
