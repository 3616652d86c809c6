"""Fast training step on flat parameter / gradient buffers.

This is what ModelPlain.optimize_parameters (reference
dlib/models/model_plain.py:318-396) becomes on MI355X: forward, MasterLoss,
backward, one finite-check flag (instead of ~660 host syncs per step,
dlib/utils/tools.py:55-63), optional data-parallel gradient all-reduce (RCCL,
side stream, one bucket per RSTB layer launched as soon as that layer's
gradients exist; replaces DDP, model_base.py:135-142) and a fused multi-tensor
Adam / SGD-Nesterov update (dlib/utils/utils_instance.py:216-247) -- with no
Python-visible tensor math in between.
"""
import os

import torch

from . import ops


class FlatParams:
    """All float parameters of a module as views of ONE flat fp32 buffer (and
    their gradients as views of another), in named_parameters() order, so the
    optimizer and the gradient all-reduce each touch one contiguous range."""

    def __init__(self, net):
        named = [(k, p) for k, p in net.named_parameters() if p.requires_grad]
        dev = named[0][1].device
        self.names = [k for k, _ in named]
        self.numel = sum(p.numel() for _, p in named)
        # pad every tensor to a multiple of 4 floats: keeps 16-B alignment for
        # the vectorised kernels
        self.offsets = {}
        off = 0
        for k, p in named:
            self.offsets[k] = off
            off += (p.numel() + 3) // 4 * 4
        self.total = off
        self.flat = torch.zeros(off, device=dev)
        self.grad = torch.zeros(off, device=dev)
        self.gviews = {}
        for k, p in named:
            o, n = self.offsets[k], p.numel()
            self.flat[o:o + n].view_as(p).copy_(p.data)
            p.data = self.flat[o:o + n].view_as(p)
            self.gviews[k] = self.grad[o:o + n].view_as(p)
            p.grad = self.gviews[k]

    def range_of(self, prefix_list):
        """[lo, hi) flat range covering every parameter whose name starts with one
        of the prefixes (they are contiguous by construction)."""
        lo, hi = None, None
        for k in self.names:
            if any(k.startswith(pf) for pf in prefix_list):
                o = self.offsets[k]
                e = o + (self.gviews[k].numel() + 3) // 4 * 4
                lo = o if lo is None else min(lo, o)
                hi = e if hi is None else max(hi, e)
        return lo, hi


class Optimizer:
    """Adam (torch semantics, L2 weight decay) or SGD momentum/Nesterov on the flat
    buffers, plus MyStepLR / MultiStepLR learning-rate rules stepped per
    iteration as the reference does (utils_trainer.py:370, lr_scheduler.py:6-35)."""

    def __init__(self, flat: FlatParams, kind="adam", lr=2e-4, betas=(0.9, 0.999), eps=1e-8, wd=0.0,
                 momentum=0.9, nesterov=True, scheduler=None):
        self.fp, self.kind, self.base_lr = flat, kind, lr
        self.betas, self.eps, self.wd = betas, eps, wd
        self.momentum, self.nesterov = momentum, nesterov
        self.scheduler = scheduler or {}
        self.step_count = 0      # optimizer steps requested (host view; skipped ones included)
        self.sched_count = 0     # scheduler.step() calls
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat) if kind == "adam" else None
        # steps actually APPLIED, kept on the device: a step skipped by the non-finite flag
        # must not advance Adam's bias correction / SGD's first-step buffer initialisation
        # (the reference skips backward and optimizer.step() for that batch only,
        # model_plain.py:344-346) -- and the host must not sync to find out
        self.applied = torch.zeros(1, dtype=torch.int32, device=flat.flat.device)
        # the learning rate as the kernels read it (device scalar: a captured step replays with the current one)
        self.lr_dev = torch.zeros(1, device=flat.flat.device)

    @property
    def lr(self):
        s = self.scheduler
        if not s:
            return self.base_lr
        if s["type"] == "MyStepLR":
            return max(self.base_lr * s["gamma"] ** (self.sched_count // s["step_size"]), s["min_lr"])
        if s["type"] == "MultiStepLR":
            k = sum(1 for m in s["milestones"] if self.sched_count >= m)
            return self.base_lr * s["gamma"] ** k
        raise NotImplementedError(s["type"])

    def push_lr(self):
        """current learning rate -> device scalar (a fill kernel with the value as its argument: no copy, no sync)."""
        self.lr_dev.fill_(self.lr)

    def step(self, gscale=1.0, skip_flag=None, host_side=True):
        """host_side=False: enqueue the kernels only (graph capture); the caller advances the host counters and
        pushes the learning rate before every replay."""
        fp = self.fp
        if host_side:
            self.step_count += 1
            self.push_lr()
        ops.optim_tick(skip_flag, self.applied)          # applied += (flag == 0), on the device
        if self.kind == "adam":
            ops.adam_step_dc(fp.flat, fp.grad, self.m, self.v, self.applied, self.lr, self.betas[0],
                             self.betas[1], self.eps, self.wd, gscale, skip_flag, self.lr_dev)
        elif self.kind == "sgd":
            ops.sgd_step_dc(fp.flat, fp.grad, self.m, self.applied, self.lr, self.momentum, self.wd,
                            self.nesterov, gscale, skip_flag, self.lr_dev)
        else:
            raise NotImplementedError(self.kind)

    def scheduler_step(self):
        self.sched_count += 1

    # ---- checkpoints: the layout of torch.optim.{Adam,SGD}.state_dict() (what the reference's save_optimizer /
    # load_optimizer write and read, dlib/models/model_base.py:203-211), so ``<iter>_optimizerG.pth`` files travel in
    # both directions.  Parameter i of the one group = the i-th trainable entry of named_parameters()
    # (utils_instance.py:216-223).  torch ignores top-level keys it does not know: 'srhip' carries what torch keeps
    # elsewhere or not at all -- the LR rule's position (the reference rebuilds its scheduler from scratch on resume,
    # model_plain.py:60-66, so its schedule restarts; with this key ours continues) and the applied-step count.
    def _views(self, flat_buf):
        fp = self.fp
        return [flat_buf[fp.offsets[k]:fp.offsets[k] + fp.gviews[k].numel()].view_as(fp.gviews[k]) for k in fp.names]

    def state_dict(self):
        """One host sync (the applied-step counter)."""
        n = len(self.fp.names)
        applied = int(self.applied.item())
        group = {'lr': self.lr, 'weight_decay': self.wd, 'maximize': False, 'foreach': None, 'differentiable': False,
                 'fused': None, 'initial_lr': self.base_lr, 'params': list(range(n))}
        state = {}
        if self.kind == "adam":
            group.update(betas=tuple(self.betas), eps=self.eps, amsgrad=False, capturable=False,
                         decoupled_weight_decay=False)
            if applied > 0:     # torch creates a parameter's state at its first step
                for i, (m, v) in enumerate(zip(self._views(self.m), self._views(self.v))):
                    state[i] = {'step': torch.tensor(float(applied)), 'exp_avg': m.detach().cpu().clone(),
                                'exp_avg_sq': v.detach().cpu().clone()}
        else:
            group.update(momentum=self.momentum, dampening=0, nesterov=self.nesterov)
            if applied > 0:
                for i, m in enumerate(self._views(self.m)):
                    state[i] = {'momentum_buffer': m.detach().cpu().clone()}
        return {'state': state, 'param_groups': [group],
                'srhip': {'kind': self.kind, 'applied': applied, 'sched_count': self.sched_count,
                          'step_count': self.step_count, 'base_lr': self.base_lr}}

    def load_state_dict(self, sd):
        groups = sd['param_groups']
        if len(groups) != 1:
            raise ValueError(f"optimizer checkpoint with {len(groups)} parameter groups; this path has one")
        g, state, n = groups[0], sd['state'], len(self.fp.names)
        if len(g['params']) != n:
            raise ValueError(f"optimizer checkpoint covers {len(g['params'])} parameters, the network has {n}")
        is_adam = 'betas' in g
        if is_adam != (self.kind == "adam"):
            raise ValueError(f"optimizer checkpoint is {'Adam' if is_adam else 'SGD'}, the run uses {self.kind}")
        if is_adam and g.get('amsgrad', False):
            raise NotImplementedError("amsgrad optimizer state")
        # hyper-parameters follow the checkpoint, as torch's load_state_dict replaces the param_groups
        self.wd = g['weight_decay']
        if is_adam:
            self.betas, self.eps = tuple(g['betas']), g['eps']
        else:
            self.momentum, self.nesterov = g['momentum'], g['nesterov']
        self.base_lr = g.get('initial_lr', g['lr'])
        keys = [g['params'][i] for i in range(n)]
        applied = 0
        self.m.zero_()
        if self.v is not None:
            self.v.zero_()
        mv, vv = self._views(self.m), (self._views(self.v) if self.v is not None else None)
        for i, k in enumerate(keys):
            st = state.get(k)
            if not st:
                continue
            if is_adam:
                if tuple(st['exp_avg'].shape) != tuple(mv[i].shape):
                    raise ValueError(f"optimizer state {k}: shape {tuple(st['exp_avg'].shape)} vs parameter "
                                     f"{self.fp.names[i]} {tuple(mv[i].shape)}")
                mv[i].copy_(st['exp_avg'])
                vv[i].copy_(st['exp_avg_sq'])
                applied = max(applied, int(float(st['step'])))
            elif st.get('momentum_buffer') is not None:
                mv[i].copy_(st['momentum_buffer'])
                applied = max(applied, 1)      # torch's SGD keeps no count: the buffer exists = past the first step
        extra = sd.get('srhip')
        if extra is not None:
            applied = int(extra['applied'])
            self.sched_count, self.step_count = int(extra['sched_count']), int(extra['step_count'])
        else:
            # a file written by torch: the reference builds a NEW scheduler on the loaded optimizer (last_epoch = -1,
            # base_lrs = the groups' initial_lr): its learning-rate rule starts over
            self.sched_count, self.step_count = 0, applied
        self.applied.fill_(applied)
        self.push_lr()


def allreduce_range(flat_grad, lo, hi, group=None, comm_stream=None):
    """Sum-all-reduce flat_grad[lo:hi] over the data-parallel group.  On the GPU
    the collective (RCCL) is enqueued on ``comm_stream`` behind an event recorded
    on the compute stream, so it overlaps the rest of backward; the optimizer
    later waits on the stream.  Averaging is folded into the optimizer's
    ``gscale``.  (CPU tensors + gloo take the same path without streams: used by
    the world_size-2 tests.)"""
    import torch.distributed as dist
    if comm_stream is None:
        dist.all_reduce(flat_grad[lo:hi], group=group)
        return
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    comm_stream.wait_event(ev)
    with torch.cuda.stream(comm_stream):
        dist.all_reduce(flat_grad[lo:hi], group=group)


def broadcast_replica_state(flat_params, buffers, group=None, src=0):
    """What wrapping the network in DDP does at construction and, for the buffers, at every forward
    (reference model_base.py:135-142: DistributedDataParallel(network, ...), default broadcast_buffers=True):
    rank ``src``'s parameters and floating-point buffers replace every other rank's, so replicas that were
    initialised differently -- a checkpoint loaded on the master only, another seed -- start equal.  One
    collective for the flat parameter buffer, one for all buffers together (integer buffers -- index tables,
    BatchNorm's num_batches_tracked -- ride as float64: exact up to 2^53)."""
    import torch.distributed as dist
    if flat_params is not None:
        dist.broadcast(flat_params, src, group=group)
    bufs = [b for b in buffers if b is not None and b.numel() > 0]
    if not bufs:
        return
    allf = all(b.dtype == torch.float32 for b in bufs)
    flat = torch.cat([b.detach().reshape(-1).to(torch.float32 if allf else torch.float64) for b in bufs])
    dist.broadcast(flat, src, group=group)
    o = 0
    for b in bufs:
        n = b.numel()
        b.copy_(flat[o:o + n].view_as(b).to(b.dtype))
        o += n


class GradReducer:
    """Bucketed data-parallel gradient exchange (replaces DDP's reducer,
    model_base.py:135-142).  ``buckets``: flat [lo, hi) ranges in the order backward
    completes them.  The engine calls ``bucket_done(i)`` from inside backward as soon
    as bucket i's last gradient kernel is enqueued; ``finish()`` reduces whatever the
    engine did not announce (an engine may announce none, some or all of its buckets)
    -- every bucket exactly ONCE per step -- and makes the compute stream wait for the
    side stream.  Works on CPU tensors + gloo without streams (world_size-2 tests)."""

    def __init__(self, flat_grad, buckets, group=None, comm_stream=None):
        self.grad, self.buckets, self.group, self.comm_stream = flat_grad, list(buckets), group, comm_stream
        self.done = [True] * len(self.buckets)
        self.log = []                       # bucket indices in the order they were reduced (tests)
        # trace = True: HIP events around every bucket (tests / profiles: does bucket i's all-reduce run beside the
        # backward kernels of the layers after it?).  events[i] = (announced on the compute stream, all-reduce start
        # and end on the side stream)
        self.trace, self.events = False, {}

    def begin(self):
        self.done = [False] * len(self.buckets)
        self.log = []
        self.events = {}

    def bucket_done(self, i):
        if self.done[i]:
            return
        self.done[i] = True
        self.log.append(i)
        lo, hi = self.buckets[i]
        if self.trace and self.comm_stream is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev[0])
            ev[1].record(self.comm_stream)
            with torch.cuda.stream(self.comm_stream):
                import torch.distributed as dist
                dist.all_reduce(self.grad[lo:hi], group=self.group)
            ev[2].record(self.comm_stream)
            self.events[i] = ev
            return
        allreduce_range(self.grad, lo, hi, self.group, self.comm_stream)

    def reduce_flag(self, flag):
        """MAX-all-reduce of the per-step non-finite flag: every replica skips (or applies)
        the same update, so they cannot diverge."""
        import torch.distributed as dist
        if self.comm_stream is None:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)

    def finish(self, flag=None):
        for i in range(len(self.buckets)):
            self.bucket_done(i)
        if flag is not None:
            self.reduce_flag(flag)
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)


class RcclComm:
    """The C-ABI's own communicator (srhip_allreduce_*, csrc/comm.hip: RCCL resolved with dlopen) -- the exchange a caller
    without PyTorch uses.  One per process, on the current device."""

    def __init__(self, rank, world, id_bytes):
        import ctypes
        assert len(id_bytes) == 128
        self._id = ctypes.create_string_buffer(bytes(id_bytes), 128)
        h = ctypes.c_void_p(0)
        ops.call("srhip_allreduce_init", ctypes.addressof(self._id), int(rank), int(world), ctypes.addressof(h))
        self.h, self.rank, self.world = h.value, rank, world

    @staticmethod
    def unique_id():
        import ctypes
        buf = ctypes.create_string_buffer(128)
        ops.call("srhip_allreduce_unique_id", ctypes.addressof(buf))
        return buf.raw

    def bucket(self, t, compute_stream, comm_stream):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        ops.call("srhip_allreduce_bucket_async", self.h, t.data_ptr(), t.numel(), compute_stream.cuda_stream,
                 comm_stream.cuda_stream)

    def flag(self, t, compute_stream, comm_stream):
        assert t.is_cuda and t.dtype == torch.int32 and t.numel() == 1
        ops.call("srhip_allreduce_flag_async", self.h, t.data_ptr(), compute_stream.cuda_stream, comm_stream.cuda_stream)

    def wait(self, comm_stream, compute_stream):
        ops.call("srhip_allreduce_wait", self.h, comm_stream.cuda_stream, compute_stream.cuda_stream)

    def close(self):
        if self.h:
            ops.call("srhip_allreduce_destroy", self.h)
            self.h = None


class CabiGradReducer(GradReducer):
    """GradReducer on the C-ABI communicator (SRHIP_COMM=cabi): same buckets, same order, same side stream; the collectives
    are enqueued by srhip_allreduce_bucket_async instead of torch.distributed.all_reduce."""

    def __init__(self, flat_grad, buckets, comm, comm_stream):
        super().__init__(flat_grad, buckets, None, comm_stream)
        self.comm = comm

    def bucket_done(self, i):
        if self.done[i]:
            return
        self.done[i] = True
        self.log.append(i)
        lo, hi = self.buckets[i]
        self.comm.bucket(self.grad[lo:hi], torch.cuda.current_stream(), self.comm_stream)

    def reduce_flag(self, flag):
        self.comm.flag(flag, torch.cuda.current_stream(), self.comm_stream)

    def finish(self, flag=None):
        for i in range(len(self.buckets)):
            self.bucket_done(i)
        if flag is not None:
            self.reduce_flag(flag)
        self.comm.wait(self.comm_stream, torch.cuda.current_stream())


def open_cabi_comm(world, group=None):
    """One RcclComm per process: rank 0 makes the id, torch.distributed (when there is more than one rank) carries it."""
    rank = 0
    idb = [RcclComm.unique_id()]
    if world > 1:
        import torch.distributed as dist
        rank = dist.get_rank(group)
        dist.broadcast_object_list(idb, src=0, group=group)
    return RcclComm(rank, world, idb[0])


class TrainStep:
    """loss_terms: sequence of ('l1', lam) | ('l2', lam) | ('ssim', lam, window) |
    ('charbonnier', lam, eps) | ('l2sum', lam) | ('grad'|'laplace'|'lv'|'norm_grad'|
    'norm_laplace'|'norm_lv', lam, norm, ksz)  (MasterLoss = their sum,
    dlib/loss/master.py:46-56)."""

    def __init__(self, net, loss_terms=(("l1", 1.0),), optimizer=None, process_group=None,
                 world_size=1, clipgrad=0.0, ema_decay=0.0):
        """clipgrad > 0: torch.nn.utils.clip_grad_norm_(max_norm=clipgrad, norm_type=2) of the (averaged) gradient in front of
        the optimizer (G_optimizer_clipgrad, model_plain.py:350-361).  ema_decay > 0: an exponential moving average of the
        weights is kept in `self.ema_flat` (layout of fp.flat) and updated behind every applied step (E_decay, netE:
        model_plain.py:46-47,393-394, model_base.py:213-219)."""
        self.net = net
        self.fp = FlatParams(net)
        net.weights_changed()
        self.loss_terms = list(loss_terms)
        self.opt = optimizer if optimizer is not None else Optimizer(self.fp)
        self.world = world_size
        self.pg = process_group
        # SRHIP_FORCE_DDP=1 takes the bucketed all-reduce path even with one rank
        # (exercises the RCCL / side-stream plumbing on a single GPU)
        self.ddp = world_size > 1 or os.environ.get("SRHIP_FORCE_DDP", "0") == "1"
        dev = self.fp.flat.device
        self.loss_buf = torch.zeros(1 + len(self.loss_terms), device=dev)
        self._sink = torch.zeros(1, device=dev)
        # flag: THIS step's non-finite indicator (cleared at the start of every step, so one bad
        # batch skips one update, model_plain.py:344-346); sticky: OR over the steps since the
        # caller last looked (ModelPlain.check_finite)
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sticky = torch.zeros(1, dtype=torch.int32, device=dev)
        self.comm_stream = torch.cuda.Stream(device=dev) if self.ddp else None
        self.buckets = self._make_buckets() if self.ddp else []
        # SRHIP_COMM=cabi: the gradient exchange through the C-ABI's own RCCL communicator (srhip_allreduce_*: what a caller
        # without PyTorch uses) instead of torch.distributed's -- same buckets, same order, same side stream
        self.comm = None
        if self.ddp and os.environ.get("SRHIP_COMM", "torch") == "cabi":
            self.comm = open_cabi_comm(world_size, self.pg)
            self.reducer = CabiGradReducer(self.fp.grad, self.buckets, self.comm, self.comm_stream)
        else:
            self.reducer = GradReducer(self.fp.grad, self.buckets, self.pg, self.comm_stream) if self.ddp else None
        # buffers that change during training (BatchNorm running statistics: MemNet) are re-broadcast from rank 0 at
        # every step, as DDP's broadcast_buffers does at every forward; constant ones only at construction
        self._live_buffers = [b for k, b in net.named_buffers() if "running_" in k or "num_batches_tracked" in k]
        if world_size > 1:
            broadcast_replica_state(self.fp.flat, list(net.buffers()), self.pg)
            net.weights_changed()
        # SwinIR.forward multiplies the input by img_range and divides the output by it (network_swinir.py:935,968; the mean
        # is zero for one channel): prepare_input does the first, the step scales y before the loss and dy behind it
        # Only a net whose forward() does that opts in (`forward_divides_by_img_range`): EDSR-LIIF / ENLCN / NLSN / ACT carry
        # an img_range too, but it parametrises MeanShift convs their forward never applies -- scaling here would train on
        # y / img_range while forward() and test() return y.
        self.inv_range = 1.0
        if getattr(net, "forward_divides_by_img_range", False):
            self.inv_range = 1.0 / float(getattr(net, "img_range", 1.) or 1.)
        self._mean_img = None
        self.dy = None
        self.clipgrad = float(clipgrad or 0.0)
        # [global L2 norm of the last step's gradient (before clipping), the coefficient it was scaled by]
        self.clip_state = torch.zeros(2, device=dev) if self.clipgrad > 0 else None
        self.ema_decay = float(ema_decay or 0.0)
        # update_E(0) at construction = a copy of the weights (model_plain.py:82-84); a checkpoint replaces it (ModelPlain.load)
        self.ema_flat = self.fp.flat.clone() if self.ema_decay > 0 else None

    def ema_state_dict(self):
        """state_dict of the reference's netE (a second define_G instance whose PARAMETERS follow the moving average; its
        buffers are never touched by update_E and stay what the constructor made: here the network's own constants)."""
        from collections import OrderedDict
        assert self.ema_flat is not None, "E_decay == 0: there is no netE"
        fp = self.fp
        out = OrderedDict()
        for k, v in self.net.state_dict().items():
            if k in fp.offsets:
                o = fp.offsets[k]
                out[k] = self.ema_flat[o:o + v.numel()].view_as(v).detach().cpu().clone()
            else:
                out[k] = v.detach().cpu().clone()
        return out

    def load_ema_state_dict(self, sd, strict=True):
        fp = self.fp
        missing = [k for k in fp.names if k not in sd]
        if missing and strict:
            raise KeyError(f"netE checkpoint misses {missing[:4]} ... ({len(missing)} parameters)")
        for k in fp.names:
            if k in sd:
                o, n = fp.offsets[k], fp.gviews[k].numel()
                self.ema_flat[o:o + n].copy_(sd[k].reshape(-1))

    def _make_buckets(self):
        """Gradient buckets in the order backward completes them (the engine names
        them by parameter prefix); each is one contiguous range of the flat
        gradient buffer."""
        return [self.fp.range_of(pf) for pf in self.net.engine.bucket_prefixes()]

    def sync_buffers(self):
        """rank 0's BatchNorm running statistics to every rank (DDP's broadcast_buffers, model_base.py:139): called at
        the start of every step, and by ModelPlain before a distributed evaluation -- the last step's rank-local
        updates must not reach the metrics."""
        if self.world > 1 and self._live_buffers:
            broadcast_replica_state(None, self._live_buffers, self.pg)

    def loss_and_grad(self, y, target, weight=None):
        """MasterLoss value(s) + d loss / d y through the fused loss kernels.  weight: the per-pixel weights of the
        target (--ppiw, dataset_dpsr.py:925-928); only L1 consumes them (dlib/loss/main.py:45-76)."""
        if self.dy is None or self.dy.shape != y.shape:
            self.dy = torch.empty_like(y)
        lb = self.loss_buf
        for i, t in enumerate(self.loss_terms):
            first = i == 0
            part = lb[1 + i:2 + i]
            if t[0] in ("l1", "l2"):
                ops.loss_l1l2(y, target, 0 if t[0] == "l1" else 1, t[1], weight if t[0] == "l1" else None, self.dy,
                              part, grad_accum=not first)
            elif t[0] == "ssim":
                ops.ssim_loss(y, target, t[2], t[1], self.dy, part, grad_accum=not first)
            elif t[0] == "charbonnier":
                ops.loss_pointwise(y, target, 2, t[1], t[2], None, self.dy, part, grad_accum=not first)
            elif t[0] == "boundpred":       # (kind, lam, eps, ELB module | t, restore_range, color_max)
                tb = float(t[3].get_t()) if hasattr(t[3], "get_t") else float(t[3])
                ops.loss_bounded(y, target, t[1], t[2], tb, float(t[5]) if t[4] else 1.0, self.dy, part,
                                 grad_accum=not first)
            elif t[0] == "w_sparsity":      # value here; its gradient is added after backward (step())
                ops.l1_sparsity(self.fp.flat, t[1], None, part)
                if first:
                    self.dy.zero_()
            elif t[0] == "kde":
                ops.loss_kde(y, target, t[1], t[2], t[3], t[4], self.dy, part, grad_accum=not first,
                             elb_t=float(t[5].get_t()) if (t[2] == 4 and len(t) > 5) else 1.0)
            elif t[0] == "hist":
                ops.loss_hist(y, target, t[1], t[2], t[3], t[4], self.dy, part, grad_accum=not first,
                              elb_t=float(t[5].get_t()) if (t[2] == 4 and len(t) > 5) else 1.0)
            elif t[0] == "local_moments":
                ops.loss_local_moments(y, target, t[1], self.dy, part, grad_accum=not first)
            elif t[0] == "l2sum":
                ops.loss_pointwise(y, target, 3, t[1], grad=self.dy, loss_out=part, grad_accum=not first)
            elif t[0].replace("norm_", "") in ops.STENCIL_OPS:
                ops.loss_stencil(y, target, t[0].replace("norm_", ""), t[1], t[2], t[3] if len(t) > 3 else 3,
                                 t[0].startswith("norm_"), self.dy, part, grad_accum=not first)
            else:
                raise NotImplementedError(t[0])
        return self.dy

    def multiscale_loss_and_grad(self, y, inter, target):
        """loss_mslaprs (reference model_plain.py:277-314): the loss of the output plus the same loss of every
        intermediate image against the bicubically resized target (align_corners=True, clamped to [0, 1]; stock
        F.interpolate, as the reference), all divided by the number of images.  L1 / L2 terms."""
        import torch.nn.functional as F
        n = len(inter) + 1.0
        for t in self.loss_terms:
            if t[0] not in ("l1", "l2"):
                raise NotImplementedError(f"multi-scale loss (intermediate outputs) with the term {t[0]!r}: l1 / l2 only")
        outs = [y] + list(inter)
        tgts = [target] + [torch.clamp(F.interpolate(target, size=t.shape[2:], mode="bicubic", align_corners=True),
                                       0.0, 1.0) for t in inter]
        if getattr(self, "_ms", None) is None or len(self._ms[0]) != len(outs) or \
                any(a.shape != b.shape for a, b in zip(self._ms[0], outs)):
            self._ms = ([torch.empty_like(o) for o in outs],
                        torch.zeros(len(outs), len(self.loss_terms), device=y.device))
        dys, parts = self._ms
        for j, (o, tg) in enumerate(zip(outs, tgts)):
            for i, t in enumerate(self.loss_terms):
                ops.loss_l1l2(o, tg.contiguous(), 0 if t[0] == "l1" else 1, t[1] / n, None, dys[j], parts[j, i:i + 1],
                              grad_accum=i > 0)
        self.loss_buf[1:1 + len(self.loss_terms)].copy_(parts.sum(0))
        return dys[0], dys[1:]

    def step_graph(self, lr_img, hr_img, weight=None):
        """The same optimisation step replayed from a hipGraph: the ~330 launches of a SwinIR step are
        captured once per (batch, shape), the RCCL bucket all-reduces of a data-parallel run included -- every buffer is
        persistent, DropPath masks are drawn on the
        device by captured generator ops, the learning rate and the applied-step counter live in device
        memory -- and replayed with ONE host call per step (the eager step costs the CPU ~8 ms of launch
        calls).  The first call for a shape runs one eager step (it creates the buffers), the second
        captures.  Results are those of step() bit for bit."""
        for t in self.loss_terms:
            # host-evaluated schedules would be frozen into the graph at capture time: the extended log barrier's t
            # (ELB.update_t(), dlib/losses/elb.py:92-122) of BoundedPrediction and of the Bhattacharyya histogram / KDE terms
            if t[0] == "boundpred" or (t[0] in ("hist", "kde") and len(t) > 5 and t[2] == 4):
                raise NotImplementedError(f"step_graph: the loss term {t[0]!r} carries a host-side schedule (ELB t); use step()")
        key = (tuple(lr_img.shape), tuple(hr_img.shape), weight is not None)
        st = getattr(self, "_graph", None)
        if st is not None and st["g"] is not None and st.get("gen") != ops.realloc_generation():
            # a persistent buffer was replaced since the capture (an evaluation forward on a larger image grew a scratch
            # buffer, another batch shape re-made an engine buffer): the graph holds freed addresses -- eager step, re-capture
            st = self._graph = None
        if st is None or st["key"] != key:
            out = self.step(lr_img, hr_img, weight=weight)    # eager: allocates every buffer of this shape
            self._graph = {"key": key, "g": None, "lr": lr_img.clone(), "hr": hr_img.clone(),
                           "w": None if weight is None else weight.clone()}
            return out
        st["lr"].copy_(lr_img)
        st["hr"].copy_(hr_img)
        if weight is not None:
            st["w"].copy_(weight)
        if st["g"] is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # under data parallelism the bucketed all-reduces are captured with the step: the side stream joins the
            # capture through the events the reducer records; thread-local error mode keeps RCCL's watchdog thread out of it
            with torch.cuda.graph(g, capture_error_mode="thread_local" if self.ddp else "global"):
                self._enqueue(st["lr"], st["hr"], None, host_side=False, weight=st["w"])
            st["g"] = g
            st["gen"] = ops.realloc_generation()
        self.opt.step_count += 1
        self.opt.push_lr()
        st["g"].replay()
        self.opt.scheduler_step()
        self.net.weights_changed()
        return self.loss_buf

    def step(self, lr_img, hr_img, dp=None, weight=None):
        """One optimisation step.  Returns the device tensor [total, term1, ...]
        (no host sync here; read it when needed).  weight: per-pixel weights of the L1 term (shape of hr_img)."""
        out = self._enqueue(lr_img, hr_img, dp, host_side=True, weight=weight)
        self.opt.scheduler_step()
        self.net.weights_changed()
        return out

    def _enqueue(self, lr_img, hr_img, dp, host_side, weight=None):
        net = self.net
        xi, h, w = net.prepare_input(lr_img)
        assert (h, w) == tuple(xi.shape[1:3]), \
            "training patches must not need padding (SwinIR: multiples of the 8x8 window)"
        self.flag.zero_()
        self.sync_buffers()
        if dp is None:
            dp = net.sample_drop_path(xi.shape[0], xi.device)
        y = net.engine.forward(xi, dp, save=True)
        mean = getattr(net, "mean", None)
        if torch.is_tensor(mean) and bool((mean != 0).any()):
            # y / img_range + mean (network_swinir.py:968; RGB only): the mean as an image, one launch with the scale
            if self._mean_img is None or self._mean_img.shape != y.shape:
                self._mean_img = mean.to(y).expand_as(y).contiguous()
            ops.axpby(y, self._mean_img, 1.0, self.inv_range)
        elif self.inv_range != 1.0:
            ops.axpby(y, y, self.inv_range, 0.0)
        inter = getattr(net.engine, "intermediate_outs", None)
        d_inter = None
        if inter:       # MSLapSRN: the trainer's multi-scale loss (model_plain.py:277-314)
            if weight is not None:
                raise NotImplementedError("per-pixel weights with the multi-scale loss")
            dy, d_inter = self.multiscale_loss_and_grad(y, inter, hr_img)
        else:
            if weight is not None:
                assert weight.shape == hr_img.shape and weight.is_contiguous(), "per-pixel weights: the target's shape"
            dy = self.loss_and_grad(y, hr_img, weight)
        hook = None
        if self.ddp:
            self.reducer.begin()
            hook = self.reducer.bucket_done
        if self.inv_range != 1.0:
            ops.axpby(dy, dy, self.inv_range, 0.0)
        # every gradient kernel OVERWRITES its tensor (the LayerNorm-affine sums are two-stage and deterministic too);
        # the memset only keeps a parameter without a gradient path (and the alignment padding) at zero
        self.fp.grad.zero_()
        if d_inter is not None:
            net.engine.backward(dy, self.fp.gviews, on_layer_done=hook, grads_zeroed=True, d_inter=d_inter)
        else:
            net.engine.backward(dy, self.fp.gviews, on_layer_done=hook, grads_zeroed=True)
        # one device flag: non-finite loss -> the optimizer kernel skips the update
        ops.nonfinite_flag(self.loss_buf, self.flag)
        if self.ddp:
            self.reducer.finish(self.flag)
        for t in self.loss_terms:            # parameter-space term: lam*sign(w) joins the (summed) gradients;
            if t[0] == "w_sparsity":         # x world because the optimizer divides the all-reduced sum by it
                ops.l1_sparsity(self.fp.flat, t[1] * self.world, self.fp.grad, self._sink)
        torch.maximum(self.sticky, self.flag, out=self.sticky)
        if self.clipgrad > 0:                # on the averaged gradient, as the reference clips behind DDP's all-reduce
            ops.grad_norm_clip(self.fp.grad, 1.0 / self.world, self.clipgrad, self.clip_state)
        self.opt.step(gscale=1.0 / self.world, skip_flag=self.flag, host_side=host_side)
        if self.ema_flat is not None:
            ops.ema_update(self.ema_flat, self.fp.flat, self.ema_decay, self.flag)
        return self.loss_buf

    def loss_values(self):
        """[total, term1, ...] as Python floats (host sync)."""
        v = self.loss_buf[1:].tolist()
        return [sum(v)] + v
