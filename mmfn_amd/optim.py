"""AdamW over the model's flat parameter buffer, with torch.optim.AdamW's checkpoint format.

The reference trains with `optim.AdamW(model.parameters(), lr=args.lr)` (run_steps/phase2_train_net.py:256)
and saves `optimizer.state_dict()` next to the weights (`best_optim.pth`, `recent_optim.pth`, :209,216).
FusedAdamW drives mmfn_adamw_groups_f32 (one launch over all 104.8 M trained parameters, csrc/optim.hip) and reads /
writes that same state-dict layout: parameter ids in group order (one group: `model.parameters()` order), per-id
`step`, `exp_avg`, `exp_avg_sq` in checkpoint shapes (OIHW for convolutions), and no entry for the
parameters that never receive a gradient (vec/rad: the raster-map stem + layer1) -
exactly what torch's optimizer holds after a reference run.

Parameter groups: the reference also defines (and never uses) a decay / no-decay split in
`GPT.configure_optimizers` (model_vec.py:179-209).  `configure_optimizers(model)` below applies that rule to the whole
network; `FusedAdamW(model, param_groups=...)` takes such torch-style groups, each with its own lr / betas / eps /
weight_decay.  All hyper-parameters live in a small device table the kernel reads (Engine.set_hyper), so changing the
learning rate between steps costs one 512-byte copy and never re-captures a hipGraph.
"""
import torch
import torch.nn as nn

from . import params as P

_HYPER = ("lr", "betas", "eps", "weight_decay")


def configure_optimizers(model, weight_decay=0.01):
    """The reference's decay / no-decay rule (model_vec.py:179-209) over the whole model: weights of Linear / Conv2d
    modules (and the GRU cell / GAT matrices, which are plain weight matrices) decay; every bias, every LayerNorm /
    BatchNorm weight and the GPT position embeddings do not.  Returns torch-style param groups, names sorted as the
    reference sorts them."""
    decay, no_decay = set(), set()
    for mn, m in model.named_modules():
        for pn, p in m.named_parameters(recurse=False):
            fpn = "%s.%s" % (mn, pn) if mn else pn
            if pn.endswith("bias") or pn.startswith("bias"):                     # Linear/LN/BN biases, GRUCell bias_ih / bias_hh
                no_decay.add(fpn)
            elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)) or pn == "pos_emb":
                no_decay.add(fpn)
            else:
                decay.add(fpn)
    named = dict(model.named_parameters())
    assert decay | no_decay == set(named) and not decay & no_decay
    return [{"params": [named[n] for n in sorted(decay)], "weight_decay": weight_decay},
            {"params": [named[n] for n in sorted(no_decay)], "weight_decay": 0.0}]


class FusedAdamW(object):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, param_groups=None):
        self.model = model
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False)
        if param_groups is None:
            param_groups = [{"params": list(model.parameters())}]
        self.param_groups = []
        for g in param_groups:
            full = dict(self.defaults)
            full.update({k: v for k, v in g.items() if k != "params"})
            full["betas"] = tuple(full["betas"])
            full["params"] = list(g["params"])
            self.param_groups.append(full)
        by_id = {id(p): n for n, p in model.named_parameters()}
        self._group_names = []
        seen = set()
        for g in self.param_groups:
            names = []
            for p in g["params"]:
                n = by_id.get(id(p))
                if n is None:
                    raise ValueError("param_groups holds a tensor that is not a parameter of the model")
                if n in seen:
                    raise ValueError("parameter %s appears in more than one group" % n)
                seen.add(n)
                names.append(n)
            self._group_names.append(names)
        missing = set(by_id.values()) - seen
        if missing:
            raise ValueError("param_groups must cover every parameter of the model (missing e.g. %s)" % sorted(missing)[:3])
        if len(self.param_groups) > 16:
            raise ValueError("at most 16 parameter groups")
        self._install_groups()

    def _install_groups(self):
        """One group id per float4 of the flat buffer (tensors are 16-byte aligned in it), handed to the engine."""
        L = self.model._layout
        self._group_of = None
        if len(self.param_groups) > 1:
            gid = torch.zeros(L.total // 4, dtype=torch.uint8)
            for gi, names in enumerate(self._group_names):
                for n in names:
                    off, cnt = L.offsets[n]
                    gid[off // 4:(off + cnt + 3) // 4] = gi
            self._group_of = gid.to(L.device)
        if L.device.type == "cuda":
            self.model._engine_for().set_param_groups(self._group_of)

    # ------------------------------------------------------------------ stepping
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def hyper_rows(self):
        """[(lr, beta1, beta2, eps, weight_decay)] per group - what Engine.optimizer_step(groups=...) takes."""
        return [(g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"]) for g in self.param_groups]

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None  # the flat gradient buffer is overwritten, never accumulated, by every backward

    def step(self, grad_scale=1.0):
        self.model._engine_for().optimizer_step(grad_scale=grad_scale, groups=self.hyper_rows())

    # ------------------------------------------------------------------ torch.optim.AdamW-format state
    def _moment_views(self):
        """(name, exp_avg view, exp_avg_sq view) in torch's parameter-id order: group by group."""
        L = self.model._layout
        named = dict(self.model.named_parameters())
        out = []
        for names in self._group_names:
            for name in names:
                off, n = L.offsets[name]
                p = named[name]
                if P._is_conv_weight(p):
                    o, i, kh, kw = p.shape
                    view = lambda flat, off=off, n=n, o=o, i=i, kh=kh, kw=kw: flat[off:off + n].view(o, kh, kw, i).permute(0, 3, 1, 2)
                else:
                    view = lambda flat, off=off, n=n, shp=p.shape: flat[off:off + n].view(shp)
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
        groups = []
        start = 0
        for g, names in zip(self.param_groups, self._group_names):
            d = {k: v for k, v in g.items() if k != "params"}
            d.update(maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
            d["params"] = list(range(start, start + len(names)))
            start += len(names)
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        L = self.model._layout
        eng = self.model._engine_for()
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("optimizer state has %d parameter groups, this optimizer %d" % (len(sd["param_groups"]), len(self.param_groups)))
        for mine, names, theirs in zip(self.param_groups, self._group_names, sd["param_groups"]):
            if len(theirs["params"]) != len(names):
                raise ValueError("optimizer state holds %d parameters in a group, the model has %d" % (len(theirs["params"]), len(names)))
            for k in _HYPER:
                if k in theirs:
                    mine[k] = tuple(theirs[k]) if k == "betas" else theirs[k]
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
