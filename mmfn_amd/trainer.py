"""Epoch loop, validation and checkpoint / resume around the HIP training step.

Restates the reference's `Engine` (run_steps/phase2_train_net.py:44-220) and the resume block of its
`main` (:288-302).  Same attributes (`cur_epoch, cur_iter, bestval, bestval_epoch, train_loss, val_loss`),
same files in the log directory:
    recent.log        JSON {epoch, iter, bestval, bestval_epoch, train_loss, val_loss}        (:193-204)
    best_model.pth    model.state_dict() when the validation loss improved                    (:207-211)
    best_optim.pth    optimizer.state_dict() (torch.optim.AdamW layout)                       (:209)
    model.pth, recent_optim.pth   always                                                      (:215-216)
so a run can be resumed by either implementation.  Differences, all on the host side:
  * the step is the fused one (forward + L1 + backward + bucketed all-reduce + AdamW on device buffers);
    `fused=False` runs the reference's literal sequence through autograd instead;
  * the loss is accumulated on the device and read back once per `log_every` steps, not every step
    (:109 forces a host sync per step; anomaly mode :107 is not reproduced);
  * checkpoints always carry un-prefixed keys (the reference mixes `module.`-prefixed and plain keys
    under DDP, :208 vs :216).
"""
import json
import os

import torch

from . import data as D
from . import ops
from .parallel import StaticBatchStep, StaticEvalStep


def _evict_lru(cache):
    """Drop the least recently used entry of a capture cache (dict in LRU order) and free it - hipGraph exec, its private
    pool, the static input copies - HERE: nothing may still be replaying it, so wait for the device first.  The entry is popped
    inside this function and never handed in as an argument: a caller's argument slot would keep the capture alive until this
    call returned, i.e. past gc.collect() / drain_graveyard().  The "seen" / "eager" placeholders hold nothing."""
    state = cache.pop(next(iter(cache)))
    if isinstance(state, str) or state is None:
        return
    import gc
    from . import graphs
    torch.cuda.synchronize()
    del state                  # the last reference: Graph.__del__ parks the capture in the graveyard now
    gc.collect()
    graphs.drain_graveyard()   # the capture's hipGraphExecs are destroyed here, with the device idle


class Trainer(object):
    def __init__(self, device, log_dir, cur_epoch=0, cur_iter=0):
        self.cur_epoch = cur_epoch
        self.cur_iter = cur_iter
        self.bestval_epoch = cur_epoch
        self.train_loss = []
        self.val_loss = []
        self.bestval = 1e10
        self.device = device
        self.logdir = log_dir
        self.max_captured_shapes = 12  # ragged batches (lane buckets, LiDAR sizes) multiply the shapes: bound the captures

    # ------------------------------------------------------------------ one epoch of training
    def train(self, model, dataloader_train, config, optimizer, dp=None, fused=True, log_every=50, on_log=None, graph=True,
              lane_bucket=16):
        """One epoch.  graph=True (fused path): the second batch of a given shape captures the step into hipGraphs over
        static input buffers and every later batch of that shape only copies its inputs and replays (the first one runs
        eagerly and sizes the buffers); lane sets are zero-padded to a multiple of `lane_bucket` lanes so that ragged
        batches fall into few shapes (padded lanes are masked by lane_num: outputs are unchanged, parameter gradients to
        the last ulp of their row sums)."""
        model.train()
        eng = model._engine_for()
        if not hasattr(self, "_static_steps"):
            self._static_steps = {}      # input-shape signature -> "seen" | "eager" | StaticBatchStep, in LRU order
        total = torch.zeros(1, dtype=torch.float32, device=model._layout.device)
        window = torch.zeros_like(total)
        num_batches = 0
        for args, gt in D.DevicePrefetcher(dataloader_train, self.device, config, variant=model.variant):
            if fused:
                inp = args if isinstance(args, dict) else model._pack(*args)  # raw-frame batches are engine inputs already
                # per-group (lr, beta1, beta2, eps, weight_decay): read every step, so an LR scheduler just works - they go
                # to the device table the AdamW kernel reads, captured graphs stay valid
                adam = dict(groups=_hyper_rows(optimizer))
                lr = optimizer.param_groups[0]["lr"]
                inp = _bucket_lanes(inp, lane_bucket)  # in both modes, so that eager and replayed steps are bit-identical
                if graph:
                    sig = StaticBatchStep.signature(inp, gt)
                    state = self._static_steps.pop(sig, None)
                    if state is None:  # first batch of this shape: eager (allocates the engine's buffers for it)
                        state = "seen"
                        loss = eng.train_step(inp, gt, lr=lr, dp=dp, **adam)
                    else:
                        if state == "seen":
                            try:
                                state = StaticBatchStep(eng, dp, inp, gt, lr, **adam)
                            except RuntimeError as exc:  # a failed capture must not take the run down: this shape stays eager
                                import warnings
                                warnings.warn("hipGraph capture of the training step failed (%s); continuing with eager launches" % exc)
                                torch.cuda.synchronize()
                                state = "eager"
                        loss = eng.train_step(inp, gt, lr=lr, dp=dp, **adam) if state == "eager" else state(inp, gt, lr=lr, **adam)
                    self._static_steps[sig] = state  # re-inserted last: the dict is the LRU order
                    while len(self._static_steps) > self.max_captured_shapes:
                        _evict_lru(self._static_steps)
                else:
                    loss = eng.train_step(inp, gt, lr=lr, dp=dp, **adam)
            else:
                if dp is not None or isinstance(args, dict):
                    raise NotImplementedError("the autograd path takes reference-format batches on one GPU; use fused=True")
                for p in model.parameters():
                    p.grad = None
                pred = model(*args)
                loss = torch.nn.functional.l1_loss(pred, gt, reduction="none").mean()
                loss.backward()
                optimizer.step()
                loss = loss.detach().view(1)
            total += loss
            window += loss
            self.cur_iter += 1
            num_batches += 1
            if on_log is not None and num_batches % log_every == 0:
                on_log({"loss": float(window.item()) / log_every, "iter": self.cur_iter})
                window.zero_()
        self.train_loss.append(float(total.item()) / max(num_batches, 1))
        self.cur_epoch += 1
        return self.train_loss[-1]

    # ------------------------------------------------------------------ validation (no grad, eval-mode BN, no dropout)
    def validate(self, model, dataloader_val, config, graph=True, lane_bucket=16):
        """Mean L1 loss over the validation set in eval mode (phase2_train_net.py:124-177).  graph=True: as in train(), the
        second batch of a shape captures the forward into a hipGraph over static inputs (parallel.StaticEvalStep) and later
        batches of that shape replay it; lane sets are padded to a multiple of `lane_bucket` (padded lanes are masked by
        lane_num: the loss is unchanged)."""
        model.eval()
        eng = model._engine_for()
        if not hasattr(self, "_static_evals"):
            self._static_evals = {}      # input-shape signature -> "seen" | "eager" | StaticEvalStep, in LRU order
        total = torch.zeros(1, dtype=torch.float32, device=model._layout.device)
        num_batches = 0
        with torch.no_grad():
            for args, gt in D.DevicePrefetcher(dataloader_val, self.device, config, variant=model.variant):
                inp = args if isinstance(args, dict) else model._pack(*args)
                if graph:
                    inp = _bucket_lanes(inp, lane_bucket)
                    sig = StaticBatchStep.signature(inp, gt)
                    state = self._static_evals.pop(sig, None)
                    if state is None:
                        state = "seen"
                        _, loss = eng.forward(inp, False, gt)
                    else:
                        if state == "seen":
                            try:
                                state = StaticEvalStep(eng, inp, gt)
                            except RuntimeError as exc:
                                import warnings
                                warnings.warn("hipGraph capture of the validation step failed (%s); continuing with eager launches" % exc)
                                torch.cuda.synchronize()
                                state = "eager"
                        loss = eng.forward(inp, False, gt)[1] if state == "eager" else state(inp, gt)
                    self._static_evals[sig] = state
                    while len(self._static_evals) > self.max_captured_shapes:
                        _evict_lru(self._static_evals)
                else:
                    _, loss = eng.forward(inp, False, gt)
                total += loss
                num_batches += 1
        if num_batches:
            self.val_loss.append(float(total.item()) / num_batches)
            return self.val_loss[-1]
        return None

    # ------------------------------------------------------------------ checkpoints
    def _log_table(self):
        return {"epoch": self.cur_epoch, "iter": self.cur_iter, "bestval": self.bestval, "bestval_epoch": self.bestval_epoch,
                "train_loss": self.train_loss, "val_loss": self.val_loss}

    def save(self, model, optimizer, logdir=None):
        logdir = logdir or self.logdir
        os.makedirs(logdir, exist_ok=True)
        best = bool(self.val_loss) and self.val_loss[-1] <= self.bestval
        if best:
            self.bestval = self.val_loss[-1]
            self.bestval_epoch = self.cur_epoch
        weights = _plain_state_dict(model)
        opt_state = optimizer.state_dict()
        # every file goes to a temporary name first and is renamed into place, so no file is ever torn; recent.log is written
        # LAST and records size + a content hash of the files it belongs to: a crash between the renames can leave model.pth one save
        # newer than recent_optim.pth, and resume() then SAYS so (the files stay plain state_dicts the reference can load, so
        # the pairing cannot be stored inside them)
        if best:
            _atomic_save(weights, os.path.join(logdir, "best_model.pth"))
            _atomic_save(opt_state, os.path.join(logdir, "best_optim.pth"))
        _atomic_save(weights, os.path.join(logdir, "model.pth"))
        _atomic_save(opt_state, os.path.join(logdir, "recent_optim.pth"))
        tmp = os.path.join(logdir, "recent.log.tmp")
        table = self._log_table()
        table["files"] = {n: _stamp(os.path.join(logdir, n)) for n in ("model.pth", "recent_optim.pth", "best_model.pth", "best_optim.pth")
                          if os.path.isfile(os.path.join(logdir, n))}
        with open(tmp, "w") as f:
            f.write(json.dumps(table))
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, os.path.join(logdir, "recent.log"))
        return best

    def resume(self, model, optimizer, logdir=None, which="best"):
        """Pick a run up from its log directory (phase2_train_net.py:288-302 loads the `best_*` pair)."""
        logdir = logdir or self.logdir
        path = os.path.join(logdir, "recent.log")
        if not os.path.isfile(path):
            return False
        with open(path) as f:
            table = json.load(f)
        self.cur_epoch = table["epoch"]
        self.cur_iter = table.get("iter", self.cur_iter)
        self.bestval = table["bestval"]
        self.bestval_epoch = table.get("bestval_epoch", self.cur_epoch)
        self.train_loss = table["train_loss"]
        self.val_loss = table["val_loss"]
        names = ("best_model.pth", "best_optim.pth") if which == "best" else ("model.pth", "recent_optim.pth")
        if not all(os.path.isfile(os.path.join(logdir, n)) for n in names):
            # no validation set / no improvement yet: save() never wrote the best_* pair - continue from the recent one
            names = ("model.pth", "recent_optim.pth")
        stale = [n for n in names if n in table.get("files", {}) and not _same_save(table["files"][n], _stamp(os.path.join(logdir, n)))]
        if stale:
            import warnings
            warnings.warn("checkpoint file(s) %s differ in size or content from what recent.log recorded (a save interrupted between "
                          "its renames, or files replaced afterwards): model, optimizer state and counters may come from different "
                          "saves" % ", ".join(stale))
        weights = torch.load(os.path.join(logdir, names[0]), map_location="cpu")
        model.load_state_dict({k[7:] if k.startswith("module.") else k: v for k, v in weights.items()})
        optimizer.load_state_dict(torch.load(os.path.join(logdir, names[1]), map_location="cpu"))
        return True


def _bucket_lanes(inp, bucket):
    lane = inp.get("lane")
    if lane is None or bucket <= 1 or lane.shape[1] % bucket == 0:
        return inp
    pad = bucket - lane.shape[1] % bucket
    out = dict(inp)
    out["lane"] = torch.nn.functional.pad(lane, (0, 0, 0, 0, 0, pad))
    return out


def _stamp(path):
    """[size, sha256 of the first and last MiB]: identifies a save by CONTENT, so copying or restoring a log directory without
    its mtimes (cp, rsync without -t, an object-store download) does not look like an interrupted save."""
    import hashlib
    size = os.stat(path).st_size
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read(1 << 20))
        if size > (2 << 20):
            f.seek(size - (1 << 20))
            h.update(f.read(1 << 20))
    return [size, h.hexdigest()]


def _same_save(logged, now):
    """A recent.log written before the stamps became content hashes holds [size, mtime_ns]: compare the size only for those
    (an int second element), so healthy older log directories resume without a spurious "files from different saves" warning."""
    if len(logged) == 2 and isinstance(logged[1], int):
        return logged[0] == now[0]
    return list(logged) == list(now)


def _atomic_save(obj, path):
    tmp = path + ".tmp"
    torch.save(obj, tmp)
    os.replace(tmp, path)


def _hyper_rows(optimizer):
    if hasattr(optimizer, "hyper_rows"):
        return optimizer.hyper_rows()
    return [(g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"]) for g in optimizer.param_groups]


def sync_resume_state(trainer, optimizer, dist, src=0):
    """After rank `src` resumed from disk: every rank takes its epoch / iteration counters, loss history and the
    optimizer hyper-parameters (lr, betas, eps, weight decay per group).  Without this the ranks would iterate different
    epoch ranges (and deadlock in the gradient all-reduce when rank 0 leaves the loop first), seed their samplers
    differently and step with different learning rates.  Weights / moments / step counter travel separately
    (DataParallel.broadcast_parameters)."""
    payload = [None]
    if dist.get_rank() == src:
        payload[0] = {"table": trainer._log_table(),
                      "groups": [{k: g[k] for k in ("lr", "betas", "eps", "weight_decay")} for g in optimizer.param_groups]}
    dist.broadcast_object_list(payload, src=src)
    st = payload[0]
    t = st["table"]
    trainer.cur_epoch, trainer.cur_iter = t["epoch"], t["iter"]
    trainer.bestval, trainer.bestval_epoch = t["bestval"], t["bestval_epoch"]
    trainer.train_loss, trainer.val_loss = list(t["train_loss"]), list(t["val_loss"])
    if len(st["groups"]) != len(optimizer.param_groups):
        raise ValueError("rank %d has %d optimizer groups, rank %d has %d" % (dist.get_rank(), len(optimizer.param_groups), src, len(st["groups"])))
    for g, new in zip(optimizer.param_groups, st["groups"]):
        g["lr"], g["betas"], g["eps"], g["weight_decay"] = new["lr"], tuple(new["betas"]), new["eps"], new["weight_decay"]


def _plain_state_dict(model):
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def fit(model, optimizer, train_loader, val_loader, config, logdir, epochs, val_every=1, save_every=1, dp=None, rank=0,
        on_log=None, dist=None):
    """The epoch loop of phase2_train_net.py:307-322: train every epoch; rank 0 validates every `val_every`
    epochs and saves every `save_every`.  Data parallel: pass `dp` (a parallel.DataParallel), or just the initialised
    torch.distributed module as `dist` - the transport is then chosen by parallel.connect: the C-ABI RCCL communicator when it
    passes its self-test on every rank (the whole step, gradient all-reduces included, replays as ONE hipGraph per batch shape),
    else torch.distributed (four graphs per step, buckets in between); a capture that fails continues eagerly."""
    if dp is None and dist is not None and dist.get_world_size() > 1:
        from .parallel import connect
        dp, _ = connect(model, dist)
        rank = dist.get_rank()
    trainer = Trainer(model._layout.device, logdir)
    if rank == 0:
        trainer.resume(model, optimizer)
    if dp is not None:
        dp.broadcast_parameters()                       # weights, BN buffers, Adam moments, step counter, RNG
        sync_resume_state(trainer, optimizer, dp.dist)  # epoch / iteration counters, loss history, lr & co
    for epoch in range(trainer.cur_epoch, epochs):
        sampler = getattr(train_loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        trainer.train(model, train_loader, config, optimizer, dp=dp, on_log=on_log if rank == 0 else None)
        if epoch % val_every == 0 and rank == 0 and val_loader is not None:
            trainer.validate(model, val_loader, config)
            if epoch % save_every == 0:
                trainer.save(model, optimizer)
    return trainer
