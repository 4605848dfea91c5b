"""Adam for models whose backbone keeps its parameters in ONE flat arena (mmskin/backbone.py): a drop-in for the
`torch.optim.Adam(model.parameters(), lr=..., weight_decay=...)` of the reference's training scripts
(/root/reference/src/scripts/benchmark/train_pad_20.py:54, stepped at :113).

Every run of parameters that are consecutive views of one fp32 CUDA storage, with gradients that are consecutive views of one
gradient storage at the same offsets (what the plan-executed backward hands out), is stepped by ONE launch of `mmskin_adam_step`
over the whole range -- moments live in two flat tensors, `state[p]['exp_avg']` / `['exp_avg_sq']` are views of them, so
`state_dict()` / `load_state_dict()` keep torch's layout.  Everything else (the head's parameters, CPU tensors, amsgrad /
maximize / capturable / tensor-lr groups) goes through torch.optim.Adam's own step.  Same update rule, same state keys."""
import torch

from . import _lib
from ._lib import call, ptr, stream

_MIN_RUN = 1 << 16     # elements: shorter runs stay with torch's multi-tensor path


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
        self._runs = {}   # (group index, first parameter id) -> dict(params, n, m, v, t, step)

    # ---- consecutive runs of one group's parameters inside a shared storage, gradients laid out alike
    @staticmethod
    def _find_runs(params):
        by_store = {}
        for p in params:
            g = p.grad
            if (g is None or not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or g.is_sparse
                    or not p.is_contiguous() or not g.is_contiguous()):
                continue
            by_store.setdefault((p.untyped_storage().data_ptr(), g.untyped_storage().data_ptr()), []).append(p)
        runs = []
        for ps in by_store.values():
            if len(ps) < 2:
                continue
            ps.sort(key=lambda q: q.storage_offset())
            cur = [ps[0]]
            for q in ps[1:]:
                a = cur[-1]
                if (q.storage_offset() == a.storage_offset() + a.numel()
                        and q.grad.storage_offset() - q.storage_offset() == a.grad.storage_offset() - a.storage_offset()):
                    cur.append(q)
                else:
                    runs.append(cur); cur = [q]
            runs.append(cur)
        out = []
        for r in runs:
            n = sum(q.numel() for q in r)
            if n >= _MIN_RUN and r[0].data_ptr() % 16 == 0 and r[0].grad.data_ptr() % 16 == 0:
                out.append(r)
        return out

    def _run_state(self, gi, run):
        key = (gi, id(run[0]), len(run))
        st = self._runs.get(key)
        n = sum(q.numel() for q in run)
        if st is not None and st["n"] == n and all(a is b for a, b in zip(st["params"], run)):
            # a load_state_dict() since the last step replaces the per-parameter tensors: fold them back into the flat moments
            if all(self.state[q].get("exp_avg") is mv for q, mv in zip(run, st["m_views"])):
                return st
        dev = run[0].device
        m = torch.zeros(n, dtype=torch.float32, device=dev)
        v = torch.zeros(n, dtype=torch.float32, device=dev)
        t, off, mviews = 0, 0, []
        step_t = torch.zeros((), dtype=torch.float32, device=dev)
        for q in run:
            k = q.numel()
            old = self.state.get(q, {})
            if "exp_avg" in old:
                m[off:off + k].copy_(old["exp_avg"].reshape(-1)); v[off:off + k].copy_(old["exp_avg_sq"].reshape(-1))
                t = max(t, int(float(old["step"])))
            mv = m[off:off + k].view(q.shape)
            self.state[q] = {"step": step_t, "exp_avg": mv, "exp_avg_sq": v[off:off + k].view(q.shape)}
            mviews.append(mv)
            off += k
        step_t.fill_(float(t))
        st = dict(params=list(run), n=n, m=m, v=v, t=t, step=step_t, m_views=mviews)
        self._runs[key] = st
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        hidden = []
        for gi, group in enumerate(self.param_groups):
            if (group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable")
                    or isinstance(group["lr"], torch.Tensor)):
                continue
            b1, b2 = group["betas"]
            for run in self._find_runs(group["params"]):
                st = self._run_state(gi, run)
                st["t"] += 1
                st["step"].add_(1.0)
                p0 = run[0]
                call("mmskin_adam_step", p0.data_ptr(), p0.grad.data_ptr(), ptr(st["m"]), ptr(st["v"]), st["n"], float(group["lr"]),
                     float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), st["t"], stream())
                torch.autograd.graph.increment_version(run)       # the library wrote the parameters: version counters as after an in-place op
                for q in run:
                    hidden.append((q, q.grad)); q.grad = None     # torch's step skips parameters without a gradient
        base = torch.optim.Adam.step
        if getattr(base, "hooked", False):   # the class-level hook wrapper of a plain torch Adam created elsewhere: this step's own wrapper runs the hooks
            base = base.__wrapped__
        try:
            base(self)
        finally:
            for q, g in hidden:
                q.grad = g
        return loss
