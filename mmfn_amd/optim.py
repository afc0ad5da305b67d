"""AdamW over the model's flat parameter buffer, with torch.optim.AdamW's checkpoint format.

The reference trains with `optim.AdamW(model.parameters(), lr=args.lr)` (run_steps/phase2_train_net.py:256)
and saves `optimizer.state_dict()` next to the weights (`best_optim.pth`, `recent_optim.pth`, :209,216).
FusedAdamW drives mmfn_adamw (one launch over all 104.8 M trained parameters, csrc/optim.hip) and reads /
writes that same state-dict layout: one param group, parameter ids in `model.parameters()` order, per-id
`step`, `exp_avg`, `exp_avg_sq` in checkpoint shapes (OIHW for convolutions), and no entry for the
parameters that never receive a gradient (vec/rad: the raster-map stem + layer1) -
exactly what torch's optimizer holds after a reference run.
"""
import torch

from . import params as P


class FusedAdamW(object):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model = model
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False)
        self.param_groups = [dict(self.defaults, params=list(model.parameters()))]

    # ------------------------------------------------------------------ stepping
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None  # the flat gradient buffer is overwritten, never accumulated, by every backward

    def step(self, grad_scale=1.0):
        g = self.param_groups[0]
        self.model._engine_for().optimizer_step(lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"],
                                                grad_scale=grad_scale)

    # ------------------------------------------------------------------ torch.optim.AdamW-format state
    def _moment_views(self):
        L = self.model._layout
        named = dict(self.model.named_parameters())
        out = []
        for name in L.names:
            off, n = L.offsets[name]
            p = named[name]
            if P._is_conv_weight(p):
                o, i, kh, kw = p.shape
                view = lambda flat: flat[off:off + n].view(o, kh, kw, i).permute(0, 3, 1, 2)
            else:
                view = lambda flat, shp=p.shape: flat[off:off + n].view(shp)
            out.append((name, view(L.exp_avg), view(L.exp_avg_sq)))
        return out

    def state_dict(self):
        L = self.model._layout
        eng = self.model._engine_for()
        steps = int(eng.step_count.item())
        state = {}
        if steps > 0:
            for idx, (name, m, v) in enumerate(self._moment_views()):
                if name in L.unused:
                    continue
                state[idx] = {"step": torch.tensor(float(steps)), "exp_avg": m.clone().contiguous(),
                              "exp_avg_sq": v.clone().contiguous()}
        g = dict(self.param_groups[0])
        g.update(maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        g["params"] = list(range(len(L.names)))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        L = self.model._layout
        eng = self.model._engine_for()
        group = sd["param_groups"][0]
        if len(group["params"]) != len(L.names):
            raise ValueError("optimizer state holds %d parameters, the model has %d" % (len(group["params"]), len(L.names)))
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in group:
                self.param_groups[0][k] = tuple(group[k]) if k == "betas" else group[k]
        steps = set()
        L.exp_avg.zero_()
        L.exp_avg_sq.zero_()
        for idx, (name, m, v) in enumerate(self._moment_views()):
            st = sd["state"].get(idx, sd["state"].get(str(idx)))
            if st is None:
                continue
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ (%s): one shared counter is kept on the device" % sorted(steps))
        eng.step_count.fill_(steps.pop() if steps else 0)
