import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_model import make, synthetic_pair, synthetic_eps, rel
from dvae_amd import ops as _ops
V = os.environ.get("V", "")
if "nopers" in V: _ops.LSTM_PERSISTENT = False
if "noarena" in V:
    _ops.SplitKArena.take = lambda self, shape, dev: torch.zeros(shape, device=dev, dtype=torch.float32)
if "E1" in V or "E2" in V:
    # arena without the Adam clear: E1 zero at the end of the step (separate fill), E2 zero at the start of the step
    from dvae_amd.optim import FlatAdam
    _st1 = FlatAdam.step
    def _step(self, grad_scale=1.0, zero_after=False, clear_extra=None):
        _st1(self, grad_scale, zero_after, None)
        if "E1" in V and clear_extra is not None:
            clear_extra[0][:int(clear_extra[1])].zero_()
    FlatAdam.step = _step
    if "E2" in V:
        _bg = _ops.SplitKArena.begin
        def _begin(self, dev):
            _bg(self, dev)
            self.buf[:max(self.need, 4)].zero_()
        _ops.SplitKArena.begin = _begin
if "E4" in V:
    _tk = _ops.SplitKArena.take
    def _take(self, shape, dev):
        self.off = (self.off + 4095) // 4096 * 4096
        return _tk(self, shape, dev)
    _ops.SplitKArena.take = _take
if "nozero" in V:
    from dvae_amd.optim import FlatAdam
    _st = FlatAdam.step
    FlatAdam.step = lambda self, grad_scale=1.0, zero_after=False, clear_extra=None: _st(self, grad_scale, False, clear_extra)
if "nofan" in V:
    import dvae_amd.model.disentangled_vae as _m
    _m.fanout = lambda x, n: (x,) * n
B, T = 4, 64
a, b = make(B, T, lr=0.0), make(B, T, lr=0.0)
b.enable_graph(True)
for i in range(4):
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 100 + i))
    eps = synthetic_eps(B, seed=200 + i)
    a.model.eps_override = eps
    b.model.eps_override = eps
    la = a.step(x1, x2, None, train=True)
    lb = b.step(x1, x2, None, train=True)
if "A" in V:
    assert b._graph is not None and b.optimizer.t == 4 and a.optimizer.t == 4
if "B" in V:
    for x, y in ((a.optimizer.exp_avg, b.optimizer.exp_avg), (a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq)):
        assert float((x - y).norm()) <= 1e-2 * float(x.norm())
if "C" in V:
    for (n1, v1), (n2, v2) in zip(a.model.named_buffers(), b.model.named_buffers()):
        if n1.endswith("num_batches_tracked"):
            assert int(v1) == int(v2) == 8
        else:
            assert float((v1 - v2).abs().max()) <= 1e-4 * max(1.0, float(v1.abs().max())), n1
if "D" in V:
    for w in (a, b):
        for lo, hi in w.optimizer._zero_ranges:
            assert float(w.optimizer.flat_g[lo:hi].abs().max()) == 0.0
    assert _ops.splitk_arena.need > 0 and float(_ops.splitk_arena.buf.abs().max()) == 0.0 and not _ops.splitk_arena.dirty
x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 150))
eps = synthetic_eps(B, seed=250)
b.model.eps_override = eps
b.optimizer.zero_grad()
b.loss_functionGVAE2(x1, x2, *b.model(x1, x2), train=True)[0].backward()
if "E" in V:
    assert float(b.optimizer.flat_g.abs().max()) > 0.0 and not b.optimizer._grads_clean
a.model.eps_override = eps
if "bfirst" in V:
    lb = b.step(x1, x2, None, train=True); la = a.step(x1, x2, None, train=True)
else:
    la, lb = a.step(x1, x2, None, train=True), b.step(x1, x2, None, train=True)
print(V or "base", " ".join(f"{rel(p,q):.0e}" for p, q in zip(la, lb)))
lb2 = b.step(x1, x2, None, train=True)
la2 = a.step(x1, x2, None, train=True)
print("   again: a-a", f"{rel(la2[0], la[0]):.0e}", "b2-a", f"{rel(lb2[0], la[0]):.0e}",
      "params equal", bool(torch.equal(a.optimizer.flat_p, b.optimizer.flat_p)),
      "arena zero", float(_ops.splitk_arena.buf.abs().max()) == 0.0, "used", _ops.splitk_arena.off, "need", _ops.splitk_arena.need)
ws = list(_ops._pers_ws.values())[0] if getattr(_ops, "_pers_ws", None) else None
print("   ptrs arena", hex(_ops.splitk_arena.buf.data_ptr()), "pers", hex(ws.data_ptr()) if ws is not None else None,
      "a.flat_p", hex(a.optimizer.flat_p.data_ptr()), "b.flat_p", hex(b.optimizer.flat_p.data_ptr()))
