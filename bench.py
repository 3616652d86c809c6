#!/usr/bin/env python3
"""Headline benchmark: SwinIR x8 (64->512, 1 channel) training patches/sec.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torchrun supplies RANK / LOCAL_RANK / WORLD_SIZE); a step is
forward + L1 loss + backward + (RCCL gradient all-reduce) + optimizer on one
synthetic batch of 8 patches per GPU (README.md:120-197 configuration: embed 180,
depths 6x4, heads 6, window 8, mlp_ratio 2, pixelshuffledirect; SGD-Nesterov).
Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around
the dominant kernel class inside the timed region; `cpu_baseline` times the
oracle (PyTorch-fp32 CPU restatement, validated against the reference) on the
host cores, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

# algorithmic work per patch, forward (SURVEY.md 8d / BASELINE.md): fwd+bwd = 3x
SWINIR_X8_GFLOP_FWD = 68.30
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0


def synth_batch(batch, scale, device, seed):
    """H = round(rand*255)/255; L = clamp(bicubic_down(H), 0, 1)."""
    g = torch.Generator().manual_seed(seed)
    hr = (torch.rand(batch, 1, 512, 512, generator=g) * 255).round() / 255
    lr = F.interpolate(hr, scale_factor=1.0 / scale, mode="bicubic").clamp(0, 1)
    return lr.to(device), hr.to(device)


def cpu_baseline(threads):
    """Oracle fwd + L1 + bwd + SGD-Nesterov on the host: ONE 64x64->512x512 patch,
    a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sr_oracle as O
    torch.set_num_threads(threads)
    cfg = O.swinir_config(drop_path_rate=0.0)
    sd = O.swinir_init_state_dict(cfg, seed=0)
    sd = {k: (v.requires_grad_(True) if v.dtype == torch.float32 and not k.endswith("attn_mask")
              else v) for k, v in sd.items()}
    lr, hr = synth_batch(1, 8, "cpu", 0)
    params = [v for v in sd.values() if v.requires_grad]
    bufs = [torch.zeros_like(p) for p in params]
    def step(first):
        for p in params:
            p.grad = None
        loss = O.loss_l1(O.swinir_forward(sd, lr, cfg), hr)
        loss.backward()
        with torch.no_grad():
            for p, b in zip(params, bufs):
                O.sgd_nesterov_step(p, p.grad, b, first, 0.01)

    step(True)                                   # untimed warm-up
    n, t0 = 0, time.perf_counter()
    while True:                                  # bounded sample: ~12 s of CPU work, 3..12 steps
        step(False)
        n += 1
        dt = time.perf_counter() - t0
        if (dt >= 12.0 and n >= 3) or n >= 12 or dt >= 60.0:
            break
    return {"value": n / dt, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"{n} steps of batch 1 (1x64x64 -> 1x512x512), fwd+L1+bwd+SGD-Nesterov, fp32, "
                      f"{dt:.1f} s after 1 warm-up step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="patches per GPU (README --batch_size 8)")
    ap.add_argument("--loss", default="l1", choices=["l1", "l2ssim"])
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "adam"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run "
                             "--nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1 or os.environ.get("SRHIP_FORCE_DDP", "0") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL
        pg = dist.group.WORLD

    from dlib.models.network_swinir import SwinIR
    from srhip import probe
    from srhip.ops import use_bx3 as ops_use_bx3
    from srhip.train import TrainStep, Optimizer, FlatParams  # noqa: F401

    torch.manual_seed(0)                      # same weights on every rank
    net = SwinIR(upscale=8, in_chans=1, img_size=64, window_size=8, depths=[6, 6, 6, 6], embed_dim=180,
                 num_heads=[6, 6, 6, 6], mlp_ratio=2, upsampler="pixelshuffledirect").to(dev).train()
    terms = [("l1", 1.0)] if args.loss == "l1" else [("l2", 1.0), ("ssim", 5.0, 19)]
    ts = TrainStep(net, terms, process_group=pg, world_size=world)
    if args.optimizer == "sgd":   # README.md:152-159
        ts.opt = Optimizer(ts.fp, "sgd", lr=0.01, momentum=0.9, nesterov=True, wd=0.0,
                           scheduler={"type": "MyStepLR", "step_size": 30, "gamma": 0.5, "min_lr": 1e-4})
    else:
        ts.opt = Optimizer(ts.fp, "adam", lr=2e-4, wd=1e-4)
    lr_img, hr_img = synth_batch(args.batch, 8, dev, seed=1000 + rank)

    def barrier():
        if pg is not None:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ts.step(lr_img, hr_img)
    barrier()
    if not args.no_roofline:
        probe.enable("gemm_nt")
    t0 = time.perf_counter()
    for i in range(args.steps):
        # HIP events around every NT-GEMM launch of every 20th step (an event is a barrier
        # packet on the stream: a probed step runs ~18 % slower, so probing all of them
        # would cost the headline number; 1 of the default 20 steps = 192 timed launches,
        # taken mid-run)
        probe.active = "gemm_nt" if (not args.no_roofline and i % 20 == 10 % max(args.steps, 1)) else None
        ts.step(lr_img, hr_img)
    barrier()
    dt = time.perf_counter() - t0
    roof = probe.collect() if not args.no_roofline else None
    probe.disable()

    if pg is not None:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = ts.loss_values()[0]
    # secondary figure, outside the timed region: forward-only patches/s of the same net and batch
    eval_pps = None
    if rank == 0 and world == 1:
        net.eval()
        with torch.no_grad():
            for _ in range(2):
                net(lr_img)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                net(lr_img)
            torch.cuda.synchronize()
            eval_pps = args.batch * 10 / (time.perf_counter() - t1)
        net.train()
    if rank == 0:
        patches = args.batch * world * args.steps
        out = {
            "metric": "train patches/sec, SwinIR x8 64->512 1ch",
            "value": patches / dt, "unit": "patches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "SwinIR x8 README config (embed 180, depths 6+6+6+6, heads 6, "
                                   "window 8, mlp 2, pixelshuffledirect), LR 1x64x64 -> HR 1x512x512, "
                                   f"fwd + {args.loss} + bwd + {args.optimizer}",
                       "global_batch": args.batch * world, "batch_per_gpu": args.batch,
                       "parallelism": f"dp{world}", "final_loss": loss,
                       "eval_patches_per_s_one_gpu": eval_pps,
                       "matmul": ("bf16x3 split MFMA: f32 operands split into 3 bf16 parts, 6 products, "
                                  "f32 accumulate (f32-accurate)") if ops_use_bx3() else "f32 MFMA"},
        }
        gflop_step = 3.0 * SWINIR_X8_GFLOP_FWD * args.batch
        out["model_flops_frac_of_f32_mfma_peak"] = \
            gflop_step / (1000.0 * dt / args.steps) / F32_MFMA_PEAK_TFLOPS
        if roof:
            out["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            # a few dozen threads is where torch's CPU kernels peak on these
            # shapes; oversubscribing a 256-thread host is 10x slower
            out["cpu_baseline"] = cpu_baseline(min(32, os.cpu_count() or 1))
        print(json.dumps(out), flush=True)
    if pg is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
