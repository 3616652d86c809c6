"""ModelPlain: the wrapper protocol the reference's trainer consumes
(dlib/models/model_plain.py:41-456, dlib/models/model_base.py:26-211), on the
fused libsrhip training step.

Kept: init_train, feed_data(dict, need_H), optimize_parameters(epoch, step),
update_learning_rate, test, set_eval_mode / set_train_mode, current_visuals,
current_log, save / save_best / load_network (raw ``netG.state_dict()`` with
``torch.save`` -- file-compatible with the reference's ``<iter>_G.pth`` /
``G-model.pth``), save_optimizer / load_optimizer / load_optimizers
(``<iter>_optimizerG.pth`` in torch.optim's state_dict layout), save_current /
load_current, loss_fn.{l_holder,n_holder,update_t}, attributes L, E, H, netG,
save_dir.

Gradient clipping (``G_optimizer_clipgrad``, model_plain.py:350-361) and the moving-average network netE
(``E_decay``, :46-47,77-86,393-394; ``<iter>_E.pth``, ``E-current_model.pth``, ``E-<best>``) run inside the fused
step (srhip_grad_norm_clip, srhip_ema_update).  Options of the reference's step that change its numbers and are
NOT implemented raise instead of being ignored: ``--amp True`` in training (autocast + GradScaler, :322-327,348,363),
``G_regularizer_orthstep / clipstep`` (:365-387).

Changed on purpose (same observable behaviour): one device flag replaces the
per-step ``isfinite(...).item()`` + ~660 ``check_corruption`` host syncs
(model_plain.py:344,396; tools.py:55-63): a non-finite loss skips the update on
the device; ``check_finite()`` reads the flag when the caller wants the
reference's log-and-exit.  DDP is replaced by bucketed RCCL all-reduce inside
TrainStep (model_base.py:135-142).
"""
import os
from collections import OrderedDict

import torch

from dlib.models.select_network import define_G
from dlib.utils.utils_instance import define_loss, optimizer_config
from srhip.train import TrainStep, Optimizer


class ModelPlain:
    def __init__(self, args):
        self.args = args
        self.opt_train = args.train if getattr(args, 'train', None) is not None else {}
        self.is_train = getattr(args, 'is_train', True)
        # model_base.py:30: <outd_backup>/<save_dir_models>; runs without those options keep everything in outd
        base = getattr(args, 'outd_backup', None) or getattr(args, 'outd', None) or '.'
        sub = getattr(args, 'save_dir_models', None)
        self.save_dir = os.path.join(base, sub) if sub else base
        dev_id = torch.cuda.current_device() if torch.cuda.is_available() else None
        if dev_id is None:
            raise RuntimeError("ModelPlain (libsrhip) needs a GPU; there is no CPU fallback")
        self.device = torch.device(f'cuda:{dev_id}')
        self.netG = define_G(args).to(self.device)
        # --amp: the reference trains and evaluates under autocast (model_plain.py:322-327); here evaluation takes the
        # reduced-precision kernels and a TRAINING step under --amp raises (optimize_parameters): it is not implemented, and
        # training f32-grade under a flag that asks for something else would be a silent change of the run
        self.netG.amp = bool(getattr(args, 'amp', False))
        for k in ('G_regularizer_orthstep', 'G_regularizer_clipstep'):
            if float(self._opt(k, 0.0) or 0.0) > 0:
                raise NotImplementedError(f"{k} > 0 (model_plain.py:365-387: orthogonal / clipping weight regularizers) is not "
                                          f"implemented on this path")
        self.schedulers = []
        self.log_dict = OrderedDict()
        self._E, self._E_pending = None, False
        self.L = self.H = self.h_per_pixel_weight = self.L_to_H = None
        self._current = None
        self.step_fn = None
        self._eval_graphs, self._weights_version = {}, 0
        self.loss_fn = None

    def _opt(self, key, default=None):
        t = self.opt_train
        if hasattr(t, 'get'):
            v = t.get(key, None)
            return default if v is None else v
        return default

    @property
    def E_decay(self):
        return float(self._opt('E_decay', 0.0) or 0.0)

    # ---------------------------------------------------------------- train setup
    def init_train(self):
        self.load()
        self.netG.train()
        self.loss_fn = define_loss(self.args)
        world = 1
        pg = None
        if getattr(self.args, 'distributed', False):
            import torch.distributed as dist
            world, pg = dist.get_world_size(), dist.group.WORLD
        self.step_fn = TrainStep(self.netG, self.loss_fn.terms(), process_group=pg, world_size=world,
                                 clipgrad=float(self._opt('G_optimizer_clipgrad', 0.0) or 0.0), ema_decay=self.E_decay)
        self.step_fn.opt = Optimizer(self.step_fn.fp, **optimizer_config(self.args))
        self.G_optimizer = self.step_fn.opt
        self.load_optimizers()
        self.load_E()
        self.log_dict = OrderedDict()

    def load_E(self):
        """model_plain.py:75-86: with E_decay > 0, netE comes from args.netG['checkpoint_path_netE'] if that file exists
        (key 'params_ema' or a raw state_dict), else it starts as a copy of netG (update_E(0); TrainStep did that)."""
        if self.E_decay <= 0 or self.step_fn is None:
            return
        netg = self.args.netG if hasattr(self.args, 'netG') else {}
        path = netg.get('checkpoint_path_netE', None) if hasattr(netg, 'get') else None
        if path and os.path.isfile(path):
            sd = torch.load(path, map_location='cpu')
            if 'params_ema' in sd:
                sd = sd['params_ema']
            self.step_fn.load_ema_state_dict(sd, strict=bool(self._opt('E_param_strict', True)))

    def load(self):
        """model_plain.py:68-73: weights from args.netG['checkpoint_path_netG'] if that file exists
        (eval.py points it at best-models/G-model.pth); 'pretrained_netG' in the train options is
        accepted as well."""
        netg = self.args.netG if hasattr(self.args, 'netG') else {}
        path = netg.get('checkpoint_path_netG', None) if hasattr(netg, 'get') else None
        if not path and hasattr(self.opt_train, 'get'):
            path = self.opt_train.get('pretrained_netG', None)
        if path and os.path.isfile(path):
            strict = self.opt_train.get('G_param_strict', True) if hasattr(self.opt_train, 'get') else True
            self.load_network(path, self.netG, strict=strict, param_key='params')

    def _reuse_optimizer(self):
        """'G_optimizer_reuse' (utils_config.py:160, default True): optimizer state is checkpointed and resumed."""
        return bool(self.opt_train.get('G_optimizer_reuse', True)) if hasattr(self.opt_train, 'get') else True

    def load_optimizers(self):
        """model_plain.py:88-93: args.netG['checkpoint_path_optimizerG'] (main.py points it at the newest
        ``<iter>_optimizerG.pth``) is loaded when it exists and G_optimizer_reuse is on."""
        netg = self.args.netG if hasattr(self.args, 'netG') else {}
        path = netg.get('checkpoint_path_optimizerG', None) if hasattr(netg, 'get') else None
        if path and os.path.isfile(path) and self._reuse_optimizer():
            self.load_optimizer(path, self.G_optimizer)

    @staticmethod
    def save_optimizer(save_dir, optimizer, optimizer_label, iter_label):
        """model_base.py:203-206: torch.save(optimizer.state_dict()) -> <iter>_<label>.pth; the dict has
        torch.optim's layout (srhip.train.Optimizer.state_dict), so either side reads the other's file."""
        os.makedirs(save_dir, exist_ok=True)
        path = os.path.join(save_dir, f'{iter_label}_{optimizer_label}.pth')
        torch.save(optimizer.state_dict(), path)
        return path

    @staticmethod
    def load_optimizer(load_path, optimizer):
        """model_base.py:208-211."""
        optimizer.load_state_dict(torch.load(load_path, map_location='cpu', weights_only=False))

    # ---------------------------------------------------------------- data
    def feed_data(self, data, need_H=True):
        self.L = data['l_im'].to(self.device, non_blocking=True)
        # SRCNN consumes the low-resolution image already interpolated to the target size (model_plain.py:184-195)
        l2h = data.get('l_to_h_img', None) if hasattr(data, 'get') else None
        self.L_to_H = None if l2h is None else l2h.to(self.device, non_blocking=True)
        self.H = data['h_im'].to(self.device, non_blocking=True) if need_H else None
        w = data.get('h_per_pixel_weight', None)
        # --ppiw (dataset_dpsr.py:925-928): consumed by the L1 term of the fused step (dlib/loss/main.py:45-76)
        self.h_per_pixel_weight = None if w is None else w.to(self.device, non_blocking=True).float().contiguous()

    # ---------------------------------------------------------------- step
    def _net_input(self):
        if getattr(self.args, 'method', None) == 'SRCNN':
            if self.L_to_H is None:
                raise KeyError("SRCNN needs the batch key 'l_to_h_img' (the interpolated low-resolution image)")
            return self.L_to_H
        return self.L

    def _train_graph_on(self):
        """The training step replayed from a hipGraph (TrainStep.step_graph: bit for bit the eager step; one host call
        instead of hundreds to thousands of launches -- GRL's step is ~16 k launches and host-bound when eager).  Default
        for the engines that say so (`train_graph_default`: the fused SwinIR / EDSR engines and the tape nets);
        `--train_graph True / False` or SRHIP_TRAIN_GRAPH=1 / 0 force it either way.  A loss term with a host-side schedule
        (ELB t) or a capture that fails falls back to the eager step for the rest of the run."""
        if getattr(self, "_train_graph_failed", False):
            return False
        env = os.environ.get("SRHIP_TRAIN_GRAPH", "")
        if env in ("0", "1"):
            return env == "1"
        arg = getattr(self.args, "train_graph", None)
        if arg is not None:
            return bool(arg)
        return bool(getattr(getattr(self.netG, "engine", None), "train_graph_default", False))

    def _capture_failed(self, why):
        """A capture the engine does not survive: eager steps for the rest of the run.  Host-side state an aborted capture
        consumed is put back (the tape nets' liveness pool, what the engine saved for a backward that never ran); under
        data parallelism EVERY rank must take the same path (a rank replaying bucket all-reduces against ranks enqueueing
        them eagerly still matches call for call, but say so)."""
        self._train_graph_failed = True
        self.step_fn._graph = None
        eng = getattr(self.netG, 'engine', None)
        pool = getattr(getattr(eng, 'bufs', None), 'pool', None)
        if pool is not None and hasattr(pool, 'reset'):
            pool.reset()
        if getattr(eng, 'saved', None) is not None:
            eng.saved = None
        rank = ''
        if getattr(self.args, 'distributed', False):
            import torch.distributed as dist
            rank = f' [rank {dist.get_rank()}]'
        print(f"[libsrhip]{rank} training-step capture failed ({why}): eager steps from here on", flush=True)

    def optimize_parameters(self, epoch: int, current_step: int):
        if getattr(self.netG, 'amp', False):
            raise NotImplementedError("--amp True in training (torch.cuda.amp autocast + GradScaler, model_plain.py:322-327,"
                                      "348,363) is not implemented on this path: the step is f32-grade only; --amp selects the "
                                      "reduced-precision kernels for evaluation (model.test / eval.py)")
        self._weights_version += 1
        done = False
        if self._train_graph_on():
            try:
                self.step_fn.step_graph(self._net_input(), self.H, weight=self.h_per_pixel_weight)
                done = True
            except NotImplementedError:       # a host-scheduled loss term: eager from here on
                self._train_graph_failed = True
            except Exception as e:            # a capture the engine does not survive (a host read, an assert, an unsupported call)
                self._capture_failed(f"{type(e).__name__}: {str(e).splitlines()[0][:120] if str(e) else ''}")
        if not done:
            self.step_fn.step(self._net_input(), self.H, weight=self.h_per_pixel_weight)
        # the engine's output buffer is persistent (overwritten by the next step) and 3-D for the 1-channel conv nets:
        # ``self.E`` hands out a [B,1,H,W] copy, as the reference's self.E is a fresh tensor -- made when somebody reads it
        # (current_visuals), not in every step (a 8 MB copy + an allocation per iteration the training loop never looks at)
        self._E, self._E_pending = None, True

    @property
    def E(self):
        if self._E is None and self._E_pending:
            y = self.netG.engine.bufs.d.get("t.y")
            self._E = None if y is None else y.reshape(y.shape[0], 1, *y.shape[-2:]).clone()
            self._E_pending = False
        return self._E

    @E.setter
    def E(self, v):
        self._E, self._E_pending = v, False

    def update_learning_rate(self):
        pass   # TrainStep steps the LR rule once per iteration (utils_trainer.py:370)

    def current_learning_rate(self):
        return self.step_fn.opt.lr

    def check_finite(self):
        """True if no non-finite loss was seen since the last call (one host sync)."""
        bad = int(self.step_fn.sticky.item())
        self.step_fn.sticky.zero_()
        return bad == 0

    def current_log(self):
        vals = self.step_fn.loss_values()
        self.log_dict['G_loss'] = vals[0]
        self.loss_fn.l_holder = [torch.tensor(v) for v in vals]
        return self.log_dict

    # ---------------------------------------------------------------- eval
    def sync_replica_buffers(self):
        """Distributed runs: rank 0's BatchNorm running statistics to every rank, so that each rank scores its shard
        with the same ones.  A COLLECTIVE: the trainer calls it where every rank arrives (utils_trainer.train_valid /
        evaluate) -- never from test() / set_eval_mode(), which the master may run alone (eval_bsize == 1,
        utils_trainer.py:382-386)."""
        if self.step_fn is not None:
            self.step_fn.sync_buffers()

    def set_eval_mode(self):
        self.netG.eval()

    def set_train_mode(self):
        self.netG.train()

    def test(self):
        self.netG.eval()
        with torch.no_grad():
            x = self._net_input()
            self.E = self._graph_forward(x) if self._eval_graph_on() else self.netG(x)
        self.netG.train()

    # ---- the evaluation forward replayed from a hipGraph (opt-in: args.eval_graph / SRHIP_EVAL_GRAPH=1).  The small
    # networks of the evaluation sweep are a few thousand short launches per forward: their rate is set by the host's
    # launch calls, which a replay does not make.  One graph per (input shape, --amp, weights version): the first forward
    # of a key runs eagerly (it creates the engine's buffers), the second is captured, later ones replay.  Any change of
    # the weights (a training step, a checkpoint load) drops the graphs: derived weight copies are made outside them.
    def _eval_graph_on(self):
        env = os.environ.get('SRHIP_EVAL_GRAPH', '')
        if env == '0':
            return False
        # a network whose forward is launch-bound at the sweep's sizes asks for the replay itself (OmniSR: ~1100 launches,
        # 16 ms of kernels in 20 ms of forward)
        return bool(getattr(self.args, 'eval_graph', False)) or env == '1' or bool(getattr(self.netG, 'eval_graph_default', False))

    def _graph_forward(self, x):
        key = (tuple(x.shape), bool(getattr(self.netG, 'amp', False)), self._weights_version)
        st = self._eval_graphs.get(key)
        if st is None:
            self._eval_graphs = {k: v for k, v in self._eval_graphs.items() if k[2] == self._weights_version}
            if len(self._eval_graphs) >= 8:        # many input shapes (whole images of a test set): no more graphs, eager
                return self.netG(x)
            self._eval_graphs[key] = {"g": None, "x": x.clone(), "y": None}
            return self.netG(x)
        from srhip import ops as _ops
        if st["g"] is not None and st.get("gen") != _ops.realloc_generation():
            st["g"], st["y"] = None, None          # a persistent buffer was replaced since the capture: re-capture
        st["x"].copy_(x)
        if st["g"] is None:
            # one eager forward in THIS mode first: what an engine prepares lazily per mode (MemNet's evaluation-time
            # BatchNorm folds are dropped by every train() call; its fp16-storage range check reads the output on the host)
            # must not happen inside the capture -- a host copy or read there is an error
            self.netG(st["x"])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["y"] = self.netG(st["x"])
            st["g"] = g
            st["gen"] = _ops.realloc_generation()
        st["g"].replay()
        return st["y"].clone()

    def current_visuals(self, need_H=True):
        out = OrderedDict()
        out['L'] = self.L.detach().float()
        out['E'] = self.E.detach().float()
        if need_H:
            out['H'] = self.H.detach().float()
        return out

    # ---------------------------------------------------------------- checkpoints
    def save_network(self, save_dir, network, network_label, iter_label):
        os.makedirs(save_dir, exist_ok=True)
        path = os.path.join(save_dir, f'{iter_label}_{network_label}.pth')
        sd = OrderedDict((k, v.detach().cpu().clone()) for k, v in network.state_dict().items())
        torch.save(sd, path)
        return path

    def save(self, iter_label):
        """model_plain.py:95-101: <iter>_G.pth and, with G_optimizer_reuse, <iter>_optimizerG.pth."""
        path = self.save_network(self.save_dir, self.netG, 'G', iter_label)
        if self.E_decay > 0 and self.step_fn is not None:                       # <iter>_E.pth (model_plain.py:97-98)
            os.makedirs(self.save_dir, exist_ok=True)
            torch.save(self.step_fn.ema_state_dict(), os.path.join(self.save_dir, f'{iter_label}_E.pth'))
        if self.step_fn is not None and self._reuse_optimizer():
            self.save_optimizer(self.save_dir, self.G_optimizer, 'optimizerG', iter_label)
        return path

    def save_network_path(self, network, path: str):
        """raw state_dict on the CPU, as model_base.py:173-181."""
        torch.save(OrderedDict((k, v.detach().cpu().clone()) for k, v in network.state_dict().items()), path)
        return path

    def save_best(self, save_dir: str, p_name_file: str = 'model.pth'):
        """model_plain.py:131-137 (called as save_best(_dir, p_name_file='model.pth'),
        utils_trainer.py:230): writes <save_dir>/G-<p_name_file>."""
        os.makedirs(save_dir, exist_ok=True)
        if self.E_decay > 0 and self.step_fn is not None:                       # E-<name> beside G-<name> (model_plain.py:136-139)
            torch.save(self.step_fn.ema_state_dict(), os.path.join(save_dir, f'E-{p_name_file}'))
        return self.save_network_path(self.netG, os.path.join(save_dir, f'G-{p_name_file}'))

    def load_network(self, load_path, network, strict=True, param_key='params'):
        self._weights_version += 1
        sd = torch.load(load_path, map_location='cpu')
        if param_key in sd:
            sd = sd[param_key]
        if strict:
            network.load_state_dict(sd, strict=True)
        else:   # positional key match of the reference's non-strict branch (model_base.py:193-200)
            cur = network.state_dict()
            for (_, v_old), k in zip(sd.items(), list(cur.keys())):
                cur[k] = v_old
            network.load_state_dict(cur, strict=True)
        network.weights_changed()   # parameters are views of the flat buffer: copy_ kept them

    def save_current(self, save_dir: str):
        """model_plain.py:103-110 (utils_trainer.py:1208): <save_dir>/G-current_model.pth."""
        os.makedirs(save_dir, exist_ok=True)
        if self.E_decay > 0 and self.step_fn is not None:
            torch.save(self.step_fn.ema_state_dict(), os.path.join(save_dir, 'E-current_model.pth'))
        return self.save_network_path(self.netG, os.path.join(save_dir, 'G-current_model.pth'))

    def load_current(self, save_dir: str):
        """model_plain.py:112-121 (utils_trainer.py:1289): loads G-current_model.pth if present."""
        path = os.path.join(save_dir, 'G-current_model.pth')
        if os.path.isfile(path):
            self.load_network(path, self.netG, strict=True, param_key='params')
        path_e = os.path.join(save_dir, 'E-current_model.pth')
        if self.E_decay > 0 and self.step_fn is not None and os.path.isfile(path_e):
            sd = torch.load(path_e, map_location='cpu')
            self.step_fn.load_ema_state_dict(sd.get('params_ema', sd), strict=True)

    def flush(self):
        self.L = self.E = self.H = None

    def info_network(self):
        n = sum(p.numel() for p in self.netG.parameters())
        return f'{self.netG.__class__.__name__}: {n} parameters'
