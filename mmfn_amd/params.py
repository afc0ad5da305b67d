"""Parameter skeleton of MMFN and its flat HBM layout.

The nn.Module tree built here carries NO compute: its only job is to own parameters/buffers under
exactly the names, shapes and registration order of the reference checkpoint
(mmfn_utils/models/model_vec.py:418-485,631-651; model_img.py:249-309; model_rad.py:419-490,
853-884; torchvision ResNet-18/34 naming), so that `state_dict()` / `load_state_dict()` /
`parameters()` interoperate with reference checkpoints and optimizers unchanged.

HBM layout (MI355X-first): all trainable parameters live in ONE flat fp32 buffer, gradients and
the two Adam moments in three more of the same shape.  That gives a single fused AdamW launch and
lets the data-parallel all-reduce run over a few large contiguous buckets.  Inside the buffer
  * conv weights are stored [Cout][KH][KW][Cin] (the K-contiguous operand of the implicit-GEMM
    kernels); the nn.Parameter is a permuted VIEW with the checkpoint's [Cout,Cin,KH,KW] shape,
  * key/query/value weights (and biases) of every attention block are adjacent, so the three
    projections run as one [3C, C] GEMM,
  * parameters that never receive a gradient in this variant (vec/rad: the raster-map stem and
    layer1, model_vec.py:430 — 21 tensors) sit at the tail, outside the optimizer's range.
"""
import math

import torch
import torch.nn as nn


# ----------------------------------------------------------------------------- containers
class Bag(nn.Module):
    """Pure parameter container."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: compute runs in mmfn_amd.engine")


def _basic_block(cin, cout, stride):
    b = Bag()
    b.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
    b.bn1 = nn.BatchNorm2d(cout)
    b.relu = nn.ReLU(inplace=True)
    b.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
    b.bn2 = nn.BatchNorm2d(cout)
    b.downsample = None
    if stride != 1 or cin != cout:
        b.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
    b.stride = stride
    return b


def resnet_trunk(depths, in_channels):
    """torchvision resnet18/34 attribute names; fc stripped as the reference does (model_vec.py:23,59)."""
    t = Bag()
    t.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
    t.bn1 = nn.BatchNorm2d(64)
    t.relu = nn.ReLU(inplace=True)
    t.maxpool = nn.MaxPool2d(3, 2, 1)
    cin = 64
    for i, (w, d) in enumerate(zip((64, 128, 256, 512), depths)):
        blocks = []
        for j in range(d):
            blocks.append(_basic_block(cin, w, 2 if (j == 0 and i > 0) else 1))
            cin = w
        setattr(t, "layer%d" % (i + 1), nn.Sequential(*blocks))
    t.avgpool = nn.AdaptiveAvgPool2d((1, 1))
    t.fc = nn.Sequential()
    for m in t.modules():  # torchvision's init
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return t


def gpt_params(c, cfg, n_modal):
    g = Bag()
    g.n_embd = c
    g.n_modal = n_modal
    g.pos_emb = nn.Parameter(torch.zeros(1, n_modal * cfg.seq_len * cfg.vert_anchors * cfg.horz_anchors, c))
    g.vel_emb = nn.Linear(1, c)
    g.drop = nn.Dropout(cfg.embd_pdrop)
    blocks = []
    for _ in range(cfg.n_layer):
        b = Bag()
        b.ln1 = nn.LayerNorm(c)
        b.ln2 = nn.LayerNorm(c)
        a = Bag()
        a.key = nn.Linear(c, c)
        a.query = nn.Linear(c, c)
        a.value = nn.Linear(c, c)
        a.attn_drop = nn.Dropout(cfg.attn_pdrop)
        a.resid_drop = nn.Dropout(cfg.resid_pdrop)
        a.proj = nn.Linear(c, c)
        a.n_head = cfg.n_head
        b.attn = a
        b.mlp = nn.Sequential(nn.Linear(c, cfg.block_exp * c), nn.ReLU(True), nn.Linear(cfg.block_exp * c, c),
                              nn.Dropout(cfg.resid_pdrop))
        blocks.append(b)
    g.blocks = nn.Sequential(*blocks)
    g.ln_f = nn.LayerNorm(c)
    for m in g.modules():  # model_vec.py:170-177
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, 0.0, 0.02)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
    return g


def vectornet_params(lane_channels=7, hidden=64, layers=3, pos_dim=64, heads=2, fusion_dim=128):
    v = Bag()
    sub = Bag()
    sub.layers = nn.Sequential()
    cin = lane_channels
    for i in range(layers):
        mlp = Bag()
        mlp.mlp = nn.Sequential(nn.Linear(cin, hidden), nn.LayerNorm(hidden), nn.ReLU())
        sub.layers.add_module("mlp_%d" % i, mlp)
        cin = 2 * hidden
    v.lane_subgraph = sub
    v.pos_emb = nn.Sequential(nn.Linear(2, pos_dim), nn.LayerNorm(pos_dim), nn.GELU(), nn.Linear(pos_dim, pos_dim))
    l2l = Bag()
    l2l.attend = nn.Softmax(dim=-1)
    l2l.to_qkv = nn.Linear(2 * hidden, 6 * hidden, bias=False)
    l2l.to_out = nn.Sequential(nn.Linear(2 * hidden, 2 * hidden), nn.Dropout(0.0))
    l2l.heads = heads
    v.L2L = l2l
    v.agent_fusion = nn.Sequential(nn.Linear(pos_dim + 2 * hidden, fusion_dim), nn.LayerNorm(fusion_dim), nn.GELU(),
                                   nn.Linear(fusion_dim, 2 * hidden))
    v.generator = nn.Sequential(nn.Linear(2 * hidden, hidden), nn.LayerNorm(hidden), nn.GELU(),
                                nn.Linear(hidden, 64 * 64 * 64))
    return v


def gat_params(nfeat, nhid, dropout, alpha, nheads):
    g = Bag()
    g.dropout, g.alpha, g.nheads = dropout, alpha, nheads
    for i in range(nheads):
        a = Bag()
        a.W = nn.Parameter(torch.zeros(nfeat, 2 * nhid))
        nn.init.xavier_normal_(a.W.data, gain=1.414)
        a.a = nn.Parameter(torch.zeros(2 * nhid, nhid))
        nn.init.xavier_normal_(a.a.data, gain=1.414)
        g.add_module("attention_%d" % i, a)
    g.mlp_1 = nn.Sequential(nn.Linear(nheads * nhid, 256), nn.Dropout(dropout))
    g.mlp_2 = nn.Sequential(nn.Linear(nheads * nhid, 128), nn.Dropout(dropout))
    g.avgpool = nn.AdaptiveAvgPool2d((1, 1))
    return g


def encoder_params(cfg, variant):
    e = Bag()
    e.avgpool = nn.AdaptiveAvgPool2d((cfg.vert_anchors, cfg.horz_anchors))
    e.image_encoder = Bag()
    e.image_encoder.normalize = True
    e.image_encoder.features = resnet_trunk((3, 4, 6, 3), 3)
    e.img_map_encoder = Bag()
    e.img_map_encoder.normalize = True
    e.img_map_encoder.features = resnet_trunk((3, 4, 6, 3), 3)
    e.lidar_encoder = Bag()
    e.lidar_encoder._model = resnet_trunk((2, 2, 2, 2), 2)
    if variant in ("vec", "rad"):
        e.vectornet_encoder = vectornet_params(lane_channels=getattr(cfg, "lane_channels", 7))
    if variant == "rad":
        e.radar_encoder = gat_params(5, cfg.hidden, cfg.attn_pdrop, cfg.alpha, cfg.nb_heads)
    n_modal = cfg.n_views + 2
    for i, c in enumerate((64, 128, 256, 512)):
        extra = 1 if (variant == "rad" and i == 3) else 0
        setattr(e, "transformer%d" % (i + 1), gpt_params(c, cfg, n_modal + extra))
    return e


# ----------------------------------------------------------------------------- flat layout
def _is_conv_weight(p):
    return p.dim() == 4


class FlatLayout(object):
    """Assigns every parameter a slice of the flat buffers and re-points .data/.grad at views."""

    def __init__(self, module, unused_names=()):
        self.module = module
        named = list(module.named_parameters())
        self.names = [n for n, _ in named]
        unused = set(unused_names)
        order = self._storage_order(named, unused)
        self.offsets = {}
        off = 0
        for name, p in order:
            if name in unused and "tail" not in self.__dict__:
                self.tail = off  # first never-trained parameter
            n = p.numel()
            self.offsets[name] = (off, n)
            off += (n + 3) // 4 * 4  # keep every tensor 16-byte aligned
        if "tail" not in self.__dict__:
            self.tail = off
        self.total = off
        # [begin, end) of each backward stage inside the trained range
        self.stage_ranges = []
        for st in range(4):
            offs = [self.offsets[n] for n, _ in order if n not in unused and self.stage_of(n) == st]
            if offs:
                self.stage_ranges.append((min(o for o, _ in offs), max((o + (k + 3) // 4 * 4) for o, k in offs)))
            else:
                self.stage_ranges.append((0, 0))
        # [begin, end) of every non-empty (stage, readiness group) inside the trained range, in storage order
        self.group_ranges = {}
        for n, _ in order:
            if n in unused:
                continue
            key = (self.stage_of(n), self.GROUPS[self.group_of(n)])
            o, k = self.offsets[n]
            b, e = self.group_ranges.get(key, (o, o))
            self.group_ranges[key] = (min(b, o), max(e, o + (k + 3) // 4 * 4))
        self.unused = unused
        self.device = None
        self.params = self.grads = self.exp_avg = self.exp_avg_sq = None
        self.buffers_flat = None

    @staticmethod
    def stage_of(name):
        """Backward completion stage of a parameter's gradient (0 finishes first): the fusion scale
        it belongs to, deepest first.  Gradient buckets for the data-parallel all-reduce are the
        contiguous stage ranges of the flat buffer, so each bucket can be reduced while the
        shallower stages are still back-propagating."""
        for s, (gpt, layer) in enumerate((("transformer4", "layer4"), ("transformer3", "layer3"),
                                           ("transformer2", "layer2"))):
            if gpt in name or layer in name:
                return s
        if name.startswith(("join.", "decoder.", "output.")) or "radar_encoder" in name:
            return 0
        return 3

    # readiness order of the gradients inside one backward stage (Engine.backward_scale): head / radar, then the fusion
    # transformer, then - in parallel lanes - VectorNet and the three ResNet trunks
    GROUPS = ("head", "gpt", "vec", "img", "lid", "map", "other")

    @classmethod
    def group_of(cls, name):
        """Readiness group of a parameter inside its backward stage (index into GROUPS): the data-parallel buckets are the
        (stage, group) ranges of the flat buffer, each reduced as soon as the engine reports it complete."""
        if name.startswith(("join.", "decoder.", "output.")) or "radar_encoder" in name:
            return 0
        for key, g in (("transformer", 1), ("vectornet_encoder", 2), ("image_encoder", 3), ("lidar_encoder", 4), ("img_map_encoder", 5)):
            if key in name:
                return g
        return 6

    def _storage_order(self, named, unused):
        """Storage order != registration order: sort by backward stage and readiness group, pack k/q/v of each attention
        block adjacently (weights, then biases) and push never-trained tensors to the tail."""
        used = sorted([(n, p) for n, p in named if n not in unused], key=lambda np_: (self.stage_of(np_[0]), self.group_of(np_[0])))
        tail = [(n, p) for n, p in named if n in unused]
        by_name = dict(used)
        taken = set()
        out = []
        for n, p in used:
            if n in taken:
                continue
            if n.endswith(".attn.key.weight"):
                base = n[:-len("key.weight")]
                group = [base + "key.weight", base + "query.weight", base + "value.weight",
                         base + "key.bias", base + "query.bias", base + "value.bias"]
                for gname in group:
                    out.append((gname, by_name[gname]))
                    taken.add(gname)
                continue
            out.append((n, p))
            taken.add(n)
        return out + tail

    def materialize(self, device):
        """(Re)allocate the flat buffers on `device` from the parameters' current values."""
        device = torch.device(device)
        params = torch.zeros(self.total, dtype=torch.float32, device=device)
        grads = torch.zeros(self.total, dtype=torch.float32, device=device)
        named = dict(self.module.named_parameters())
        self.views, self.grad_views, self.storage_views = {}, {}, {}
        for name, (off, n) in self.offsets.items():
            p = named[name]
            src = p.detach()
            if _is_conv_weight(p):
                o, i, kh, kw = p.shape
                store = params[off:off + n].view(o, kh, kw, i)
                store.copy_(src.permute(0, 2, 3, 1))
                view = store.permute(0, 3, 1, 2)
                gstore = grads[off:off + n].view(o, kh, kw, i)
                gview = gstore.permute(0, 3, 1, 2)
            else:
                store = params[off:off + n].view(p.shape)
                store.copy_(src)
                view = store
                gstore = grads[off:off + n].view(p.shape)
                gview = gstore
            p.data = view
            p.grad = None
            self.views[name] = view
            self.storage_views[name] = store          # kernel-side layout (OHWI for convs)
            self.grad_views[name] = gview             # checkpoint-shaped view of the gradient
            self.storage_views["grad:" + name] = gstore
        # BatchNorm running statistics: one flat fp32 buffer (+ int64 counters)
        bufs = [(n, b) for n, b in self.module.named_buffers() if b is not None]
        fnum = sum(b.numel() for n, b in bufs if b.dtype == torch.float32)
        inum = sum(b.numel() for n, b in bufs if b.dtype == torch.int64)
        fflat = torch.zeros(max(fnum, 1), dtype=torch.float32, device=device)
        iflat = torch.zeros(max(inum, 1), dtype=torch.int64, device=device)
        fo = io = 0
        owners = dict(self.module.named_modules())
        for n, b in bufs:
            flat, o = (fflat, fo) if b.dtype == torch.float32 else (iflat, io)
            view = flat[o:o + b.numel()].view(b.shape)
            view.copy_(b.detach())
            mod_name, _, leaf = n.rpartition(".")
            owners[mod_name]._buffers[leaf] = view
            if b.dtype == torch.float32:
                fo += b.numel()
            else:
                io += b.numel()
        self.params, self.grads = params, grads
        self.exp_avg = torch.zeros_like(params)
        self.exp_avg_sq = torch.zeros_like(params)
        self.buffers_flat, self.counters_flat = fflat, iflat
        self.device = device
        return self

    # ------------------------------------------------------------------ bf16 weight shadows (bf16 training mode)
    def make_shadows(self):
        """params16: the flat parameter buffer rounded to bf16 at the same offsets (every kernel-side weight view has a bf16 twin,
        w16()).  Refreshed once per step by the engine (ops.cast_to_bf16): the master weights stay fp32."""
        if getattr(self, "params16", None) is None or self.params16.device != self.params.device:
            self.params16 = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device)
        return self.params16

    def w16(self, name):
        off, n = self.offsets[name]
        return self.params16[off:off + n].view(self.storage_views[name].shape)

    def packed16(self, first_name, count_rows, cols):
        off, _ = self.offsets[first_name]
        return self.params16[off:off + count_rows * cols].view(count_rows, cols)

    # kernel-side accessors
    def w(self, name):
        return self.storage_views[name]

    def g(self, name):
        return self.storage_views["grad:" + name]

    def packed(self, first_name, count_rows, cols=None):
        """A contiguous [rows, cols] (or [rows]) window starting at `first_name` (k/q/v packing)."""
        off, _ = self.offsets[first_name]
        if cols is None:
            return self.params[off:off + count_rows], self.grads[off:off + count_rows]
        n = count_rows * cols
        return self.params[off:off + n].view(count_rows, cols), self.grads[off:off + n].view(count_rows, cols)

    def attach_grads(self):
        """Expose the flat gradient buffer through p.grad (autograd-compatible training loops)."""
        named = dict(self.module.named_parameters())
        for name, gv in self.grad_views.items():
            named[name].grad = None if name in self.unused else gv


def default_unused(variant):
    """vec/rad never run the raster-map stem and layer1 (model_vec.py:524,540): their 21 tensors get no
    gradient and AdamW leaves them untouched."""
    if variant == "img":
        return ()
    names = ["conv1.weight", "bn1.weight", "bn1.bias"]
    for i in range(3):
        for leaf in ("conv1.weight", "bn1.weight", "bn1.bias", "conv2.weight", "bn2.weight", "bn2.bias"):
            names.append("layer1.%d.%s" % (i, leaf))
    return tuple("encoder.img_map_encoder.features." + n for n in names)
