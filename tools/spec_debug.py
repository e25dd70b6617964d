import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ava_amd import synthetic as syn, spec as sp
from oracle import spec_oracle as so
p = dict(syn.FINCH_PARAMS); p['max_dur'] = 0.3
fs = p['fs']
audio = syn.recordings(n_files=1, fs=fs, seconds=1.0)[0][0]
for kw in ({}, {"remove_dc_offset": False}, {"target_freqs": np.linspace(500.0, 9000.0, 64)}, {"fill_value": 4.0}, {"max_dur": 0.5}):
    for stretch in (False, True):
        p['time_stretch'] = stretch
        got, _ = sp.get_spec(0.21, 0.39, audio, p, fs=fs, **kw)
        want, _ = so.get_spec(0.21, 0.39, audio, p, fs=fs, **kw)
        d = np.abs(got - want)
        i = np.unravel_index(np.argmax(d), d.shape)
        print(sorted(kw), stretch, d.max(), i, got[i], want[i], (d > 6e-8).sum())
