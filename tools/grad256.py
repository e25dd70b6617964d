import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from gpu_util import build_model
from ava_amd import synthetic as syn, layout
from oracle import vae_oracle as O
B,z=256,32
x=torch.from_numpy(syn.spectrograms(B, salt=4242)); ew,ed=syn.noise(B,z,5,6)
model=build_model(z); model.noise_source=lambda b,zz:(ew,ed)
loss=model.forward(x.cuda()); loss.backward()
g=model._grads.cpu().double()
P=O.to_params(syn.fixture_parameters(z), requires_grad=True)
out=O.forward(P,x,torch.from_numpy(ew),torch.from_numpy(ed),None,True); out['loss'].backward()
offs,total=layout.arena_offsets(z)
ref=torch.zeros(total,dtype=torch.float64)
worst=0
for s in layout.param_specs(z):
    r=P[s.name].grad.reshape(-1).double(); ref[offs[s.name]:offs[s.name]+s.numel]=r
    gg=g[offs[s.name]:offs[s.name]+s.numel]
    e=float((gg-r).norm()/max(float(r.norm()),1e-30)); nr=abs(float(gg.norm())-float(r.norm()))/float(r.norm())
    worst=max(worst,e)
    if e>1e-4: print(s.name, 'rel l2 err %.2e  norm err %.2e'%(e,nr))
mask=ref!=0
print('global rel l2', float((g-ref).norm()/ref.norm()), 'worst tensor', worst, 'loss rel', abs(float(loss)-float(out['loss']))/float(out['loss']))
