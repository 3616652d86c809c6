"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol
include/srhip.h declares, host logic (flat parameters, LR rules, registry,
state_dict layout, loud failure without a GPU) and the data-parallel gradient
path on world_size-2 gloo."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    from srhip import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 38
    for name in protos:
        assert hasattr(_lib.lib, name), name
    assert _lib.lib.srhip_abi_version() >= 3
    assert isinstance(_lib.lib.srhip_last_error(), bytes)


def test_plan_queries_need_no_gpu():
    from srhip import ops
    S, n = ops.tn_plan(32768, 180, 180)
    assert S >= 1 and n == S * 180 * 180
    S9, n9 = ops.tn_plan(32768, 180, 180, conv=True)
    assert n9 == S9 * 9 * 180 * 180


def test_cpu_tensors_fail_loudly():
    from srhip import ops
    from dlib.models.network_swinir import SwinIR
    from dlib.models.network_edsr_liif import EDSR_LIIF
    from dlib import loss, metrics
    with pytest.raises(RuntimeError):
        ops.gemm_nt(torch.zeros(8, 8), torch.zeros(8, 8))
    net = SwinIR(upscale=8, in_chans=1, img_size=16, window_size=8, depths=[2], embed_dim=60,
                 num_heads=[6], mlp_ratio=2, upsampler="pixelshuffledirect")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(1, 1, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        EDSR_LIIF(scale=2, n_resblocks=1, n_feats=16)(torch.rand(1, 1, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        loss.L1(cuda_id="cpu")(epoch=0, y_pred=torch.rand(1, 1, 8, 8), y_target=torch.rand(1, 1, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        metrics.mbatch_gpu_calculate_psnr(torch.rand(1, 1, 32, 32), torch.rand(1, 1, 32, 32))
    with pytest.raises(NotImplementedError):
        SwinIR(upscale=4, in_chans=3, img_size=64, window_size=8, upsampler="pixelshuffle")


def test_state_dict_layouts_match_reference():
    from dlib.models.network_swinir import SwinIR
    from dlib.models.network_edsr_liif import EDSR_LIIF
    z = np.load(os.path.join(G, "g4_swinir_readme.npz"))
    net = SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
                 num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect")
    sd = net.state_dict()
    assert list(sd.keys()) == list(z["keys"]) and len(sd) == 366
    assert [str(tuple(v.shape)) for v in sd.values()] == list(z["shapes"])
    e = np.load(os.path.join(G, "g2_edsr_x4.npz"))
    ref_keys = [k[3:] for k in e.files if k.startswith("sd/")]
    s, nb, nf = [int(v) for v in e["cfg"]]
    assert list(EDSR_LIIF(scale=s, n_resblocks=nb, n_feats=nf).state_dict().keys()) == ref_keys


def test_registry_and_factories():
    from dlib.models.select_network import define_G
    from dlib.utils import constants
    from dlib.utils.utils_init_default_args import init_net_g
    from dlib.utils.utils_instance import define_loss, optimizer_config

    class A(dict):
        __getattr__ = dict.get
    netG = init_net_g({'net_type': constants.SWINIR}, {'scale': 8, 'n_channels': 1, 'h_size': 512})
    netG.update(swinir_depths=[6, 6, 6, 6], swinir_num_heads=[6, 6, 6, 6],
                swinir_upsampler=constants.US_PIXEL_SHUFFLE_DIRECT)
    args = A(netG=netG, train={'l2': True, 'ssim': True, 'ssim_lambda': 5.0, 'ssim_window_s': 19,
                               'G_optimizer_type': 'sgd', 'G_optimizer_lr': 0.01,
                               'G_scheduler_type': 'MyStepLR', 'G_scheduler_step_size': 30,
                               'G_scheduler_gamma': 0.5})
    net = define_G(args)
    assert sum(p.numel() for p in net.parameters()) == 7865884
    m = define_loss(args)
    assert m.n_holder == ['master_loss', 'l2', 'negative_ssim']
    assert m.terms() == [('l2', 1.0), ('ssim', 5.0, 19)]
    cfg = optimizer_config(args)
    assert cfg['kind'] == 'sgd' and cfg['scheduler']['type'] == 'MyStepLR'
    e = init_net_g({'net_type': constants.EDSR_LIIF}, {'scale': 4, 'n_channels': 1, 'h_size': 512})
    assert define_G(A(netG=e)).n_resblocks == 16
    with pytest.raises(NotImplementedError):
        define_G(A(netG={'net_type': 'NLSN'}))
    assert constants.NETTYPE_METHOD[constants.SWINIR] == 'SWINIR'


def test_flat_params_and_lr_rules():
    from srhip.train import FlatParams, Optimizer
    from dlib.learning.lr_scheduler import MyStepLR
    net = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 7))
    before = [p.detach().clone() for p in net.parameters()]
    fp = FlatParams(net)
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p, b) and p.data_ptr() >= fp.flat.data_ptr()
        assert p.grad is not None and p.grad.shape == p.shape
    assert fp.total % 4 == 0 and all(o % 4 == 0 for o in fp.offsets.values())
    lo, hi = fp.range_of(["1."])
    assert lo == fp.offsets["1.weight"] and hi == fp.total
    g = np.load(os.path.join(G, "g8_optim.npz"))
    opt = Optimizer(fp, "sgd", lr=0.01, scheduler={"type": "MyStepLR", "step_size": 30, "gamma": 0.5,
                                                  "min_lr": 1e-4})
    tp = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.SGD([tp], lr=0.01)
    sch = MyStepLR(topt, step_size=30, gamma=0.5, min_lr=1e-4)
    for it in range(300):
        opt.scheduler_step()
        topt.step()
        sch.step()
        assert abs(opt.lr - g["mysteplr"][it]) < 1e-15
        assert abs(topt.param_groups[0]["lr"] - g["mysteplr"][it]) < 1e-15


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "sr-caco-2_amd"))
from srhip.train import FlatParams, allreduce_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 2))
fp = FlatParams(net)
buckets = [fp.range_of(["2."]), fp.range_of(["1."]), fp.range_of(["0."])]   # backward order
x = torch.randn(8, 6, generator=torch.Generator().manual_seed(100 + rank))
full = torch.cat([torch.randn(8, 6, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)])
net(x).pow(2).mean().backward()                       # autograd writes into the flat views
assert all(p.grad.data_ptr() == fp.gviews[k].data_ptr() for k, p in net.named_parameters())
for lo, hi in buckets:
    allreduce_range(fp.grad, lo, hi)
fp.grad.mul_(1.0 / world)                              # the optimizer's gscale
ref = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 2))
ref.load_state_dict(net.state_dict())
ref(full).pow(2).mean().backward()                    # 1-rank large batch
for (k, p), q in zip(net.named_parameters(), ref.parameters()):
    assert torch.allclose(fp.gviews[k], q.grad, atol=1e-6), k
covered = sorted(buckets)
assert covered[0][0] == 0 and covered[-1][1] == fp.total
assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_data_parallel_gradient_path_gloo_world2(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
               OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_main_cli_contract():
    sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
    import main as M
    a = M.parse_input("--task super-resolution --scale 8 --method SWINIR --net_type swinir --n_channels 1 "
                      "--h_size 512 --batch_size 8 --G_optimizer_type sgd --G_optimizer_lr 0.01 "
                      "--swinir_depths 6+6+6+6 --swinir_embed_dim 180 --swinir_num_heads 6+6+6+6 "
                      "--swinir_mlp_ratio 2 --swinir_upsampler pixelshuffledirect --l1 False --l2 True "
                      "--ssim True --ssim_lambda 5.0 --ssim_window_s 19 --valid_n_samples 128".split())
    assert a.netG['swinir_depths'] == [6, 6, 6, 6] and a.netG['swinir_img_size'] == 64
    assert a.train['ssim_window_s'] == 19 and a.train['G_optimizer_type'] == 'sgd' and not a.train['l1']
    with pytest.raises(ValueError):
        M.parse_input("--net_type swinir --method EDSR_LIIF".split())
    with pytest.raises(NotImplementedError):
        M.parse_input("--net_type NLSN --method NLSN".split())
    e = M.parse_input("--net_type EDSR_LIIF --method EDSR_LIIF --scale 4 --h_size 512".split())
    assert e.netG['EDSR_LIIF_n_resblocks'] == 16 and e.netG['EDSR_LIIF_upscale'] == 4
