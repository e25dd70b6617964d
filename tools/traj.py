import sys, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from gpu_util import build_model, rel
from ava_amd import synthetic as syn
from oracle import vae_oracle as O
B,z,steps=16,32,12
xs=[torch.from_numpy(syn.spectrograms(B,salt=77+i)) for i in range(3)]
model=build_model(z); ew,ed=syn.noise(B,z,2002,3003); model.noise_source=lambda b,zz:(ew,ed)
P=O.to_params(syn.fixture_parameters(z),requires_grad=True); running=O.fresh_running_stats(); opt={"step":0,"m":{},"v":{}}
for s in range(steps):
    x=xs[s%3]; model.optimizer.zero_grad(); l=model.forward(x); l.backward(); model.optimizer.step()
    w,_,_=O.train_step(P,x,torch.from_numpy(ew),torch.from_numpy(ed),running,opt)
    print(s, float(l.item()), w, '%.2e'%rel(float(l.item()),w))
