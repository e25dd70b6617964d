"""Adam over the flat parameter arena, with torch.optim.Adam's state-dict layout.

Replaces ``torch.optim.Adam(self.parameters(), lr)`` of the reference
(``ava/models/vae.py:119``; update rule ``torch/optim/adam.py:414-547``, defaults
betas=(0.9,0.999), eps=1e-8, no weight decay / amsgrad).  One HIP kernel updates
all 80 tensors (``ava_adam_flat``).  ``state_dict()`` / ``load_state_dict()`` are
torch's own, so checkpoints carry ``{'state': {i: {'step','exp_avg','exp_avg_sq'}},
'param_groups': [...]}`` with i = 0..79 in ``named_parameters()`` order
 and load into the reference unchanged.
"""
import torch

from . import _lib


class FlatAdam(torch.optim.Adam):
    def __init__(self, model, lr=1e-3):
        super().__init__(list(model.parameters()), lr=lr)
        self._step_count_flat = 0
        self.rebind(model)

    def rebind(self, model):
        """(Re)attach to the model's arenas (after construction or a device move)."""
        self._model = model
        names = [n for n, _ in model.named_parameters()]
        self.param_groups[0]['params'] = [p for _, p in model.named_parameters()]
        self._names = names
        if self._step_count_flat > 0:
            self._materialise_state()

    def _views(self, name):
        o, n, shape = self._model._arena_views[name]
        return self._model._exp_avg[o:o + n].view(shape), self._model._exp_avg_sq[o:o + n].view(shape)

    def _materialise_state(self):
        """Expose exp_avg / exp_avg_sq / step per parameter exactly like torch's Adam does
        (views into the flat arenas; ``step`` is a float32 scalar tensor)."""
        self.state.clear()
        for name, p in zip(self._names, self.param_groups[0]['params']):
            m, v = self._views(name)
            self.state[p] = {'step': torch.tensor(float(self._step_count_flat), dtype=torch.float32),
                             'exp_avg': m, 'exp_avg_sq': v}

    def zero_grad(self, set_to_none=True):
        """torch semantics without a pass over the 70 MB arena: the call only records that the gradients are gone
        (``vae.py:348``); the next backward then OVERWRITES the arena, whereas a backward that follows another
        backward with no ``zero_grad()`` in between accumulates into it (``VAE._backward_device``), and ``step()``
        with no gradient since the last ``zero_grad()`` does nothing -- exactly what ``torch.optim.Adam`` does with
        ``p.grad is None`` parameters.  ``p.grad`` itself keeps pointing at its arena view (stale values until the
        next backward) instead of becoming ``None``; with ``set_to_none=False`` the arena is zeroed."""
        if getattr(self._model, "_pending_comm", None):
            self._model._finish_comm()          # a deferred all-reduce nobody consumed must not land in a cleared arena
        self._model._grad_state = "none"
        if not set_to_none:
            self._model._grads.zero_()
        return None

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        if g['weight_decay'] != 0 or g['amsgrad'] or g['maximize']:
            raise NotImplementedError("FlatAdam implements the reference's plain Adam only")
        m = self._model
        if getattr(m, "_grad_state", "filled") == "none":
            return None                         # every p.grad is "None": torch's Adam skips such parameters
        self._step_count_flat += 1
        b1, b2 = g['betas']
        lib = _lib.load()
        from . import dist as _dist
        if m._handle is not None and _dist.active():
            # Data parallel: one guarded launch per gradient bucket, each behind ITS all-reduce only -- bucket 0's update
            # runs while bucket 1 is still on the wire (the buckets arrive deferred from VAE._backward_device(defer_comm=True);
            # otherwise they are complete already and the waits are no-ops).  Elementwise the same kernel as the flat step:
            # bit-identical parameters and moments (tests/test_cpu_dist.py, tests/test_gpu_dist.py).
            # Sharded form (dist.sharded_adam): this rank's slice of every bucket, then the parameters to everyone.
            sharded = m._sharded_adam()
            pending = {b: h for b, h in m._take_pending_comm()}
            # every work handle stays alive until the step is complete (a handle owns the staging buffers of its collective)
            alive = list(pending.values())
            if None in pending:
                _dist.wait_all([pending.pop(None)])       # the status word's MAX: the kernel's guard reads it
            evs = getattr(m, "_comm_events", None)
            rc, handles, blocked = 0, [], []
            for off, cnt in m._buckets():
                h = pending.pop((off, cnt), None)
                if h is not None:
                    if evs is not None:
                        e0 = torch.cuda.Event(enable_timing=True); e0.record()
                    _dist.wait_all([h])
                    if evs is not None:
                        e1 = torch.cuda.Event(enable_timing=True); e1.record()
                        blocked.append((e0, e1))
                so, sc = _dist.shard_of(off, cnt) if sharded else (off, cnt)
                rc = lib.ava_adam_step_range(m._handle, so, sc, float(g['lr']), float(b1), float(b2),
                                             float(g['eps']), self._step_count_flat, _lib.stream())
                if rc != 0:
                    break                                 # no further collective is enqueued behind a failed launch
                if sharded:
                    handles.append(_dist.all_gather_bucket_async(m._params, off, cnt))
            _dist.wait_all(list(pending.values()) + handles)
            del alive
            if evs is not None and blocked:
                evs.extend(blocked)                       # bench.py: exposed communication = the sum of these waits per step
            if sharded:
                m._adam_state_stale = True                # exp_avg / exp_avg_sq complete on this rank for its own slices only
        elif m._handle is not None:
            rc = lib.ava_adam_step(m._handle, float(g['lr']), float(b1), float(b2), float(g['eps']),
                                   self._step_count_flat, _lib.stream())
        else:
            rc = lib.ava_adam_flat(m._params.data_ptr(), m._grads.data_ptr(), m._exp_avg.data_ptr(),
                                   m._exp_avg_sq.data_ptr(), m._params.numel(), float(g['lr']), float(b1), float(b2),
                                   float(g['eps']), self._step_count_flat, _lib.stream())
        _lib.check(rc, "Adam step")
        return None

    def state_dict(self):
        if getattr(self._model, "_adam_state_stale", False):
            # sharded optimizer: the moments of the slices other ranks own are stale here; gathering is a collective that
            # every rank has to enter, so it cannot be done silently from a rank-0-only save_state()
            raise RuntimeError("optimizer state is sharded over the ranks: call model.gather_adam_state() on EVERY rank "
                               "before state_dict() / save_state() (train_loop does)")
        if self._step_count_flat > 0:
            self._materialise_state()
        sd = super().state_dict()
        # detach from the arenas so that a saved checkpoint is a plain snapshot
        for st in sd['state'].values():
            st['exp_avg'] = st['exp_avg'].detach().clone()
            st['exp_avg_sq'] = st['exp_avg_sq'].detach().clone()
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [float(st['step']) for st in self.state.values() if 'step' in st]
        self._step_count_flat = int(round(max(steps))) if steps else 0
        if self._step_count_flat > 0:
            for name, p in zip(self._names, self.param_groups[0]['params']):
                st = self.state.get(p)
                if st is None:
                    continue
                m, v = self._views(name)
                m.copy_(st['exp_avg'])
                v.copy_(st['exp_avg_sq'])
            self._materialise_state()
        else:
            self._model._exp_avg.zero_()
            self._model._exp_avg_sq.zero_()
