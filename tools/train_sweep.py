#!/usr/bin/env python3
"""Training sweep over the sixteen registry networks: one MI355X, B = 8 synthetic 512 x 512 HR patches (x8: 64 x 64 LR), the
registry's default options, ModelPlain.optimize_parameters (forward + L1 + backward + Adam) -- ms per step, patches/s and the
loss before / after the timed steps.

    python tools/train_sweep.py [--batch 8] [--scale 8] [--steps 4] [--nets GRL,ACT] [--out profiles/r04_train_sweep.json]

Every network is built through the same ``main.parse_input`` / ``define_model`` path as ``main.py``; each one runs in its own
process (a tape net keeps every activation of its step in persistent buffers: tens of GiB that die with the process)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
NETS = [("swinir", "SWINIR"), ("EDSR_LIIF", "EDSR_LIIF"), ("VDSR", "VDSR"), ("DRRN", "DRRN"), ("SRCNN", "SRCNN"),
        ("MSLapSRN", "MSLAPSR"), ("MemNet", "MemNet"), ("DBPN", "DBPN"), ("SRFBN", "SRFBN"), ("ProSR", "PROSR"), ("ENLCN", "ENLCN"),
        ("NLSN", "NLSN"), ("DFCAN", "DFCAN"), ("ACT", "ACT"), ("OmniSR", "OmniSR"), ("GRL", "GRL")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--scale", type=int, default=8)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--out", default=None)
    ap.add_argument("--nets", default=None)
    ap.add_argument("--child", action="store_true", help="(internal) run the named nets in this process")
    a = ap.parse_args()
    only = set(a.nets.split(",")) if a.nets else None
    if not a.child:             # the parent never touches the GPU: one child process per network
        import subprocess
        rows = []
        for net_type, _ in NETS:
            if only is not None and net_type not in only:
                continue
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--nets", net_type, "--batch", str(a.batch), "--scale",
                   str(a.scale), "--steps", str(a.steps)]
            try:
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
                line = [l for l in p.stdout.splitlines() if l.startswith("{")]
                row = json.loads(line[-1]) if line else {"net_type": net_type, "error": (p.stderr or "no output")[-300:]}
            except subprocess.TimeoutExpired:
                row = {"net_type": net_type, "error": "timeout"}
            rows.append(row)
            print(json.dumps(row), flush=True)
            if a.out:           # rewritten after every row: a sweep cut short keeps what it measured
                with open(a.out, "w") as f:
                    json.dump({"what": "ModelPlain.optimize_parameters (forward + L1 + backward + Adam, the product's default: one hipGraph replay per step where the engine offers it, eager otherwise) on "
                               "synthetic 512x512 HR patches, one MI355X; registry default options per network; one process "
                               "per network", "rows": rows}, f, indent=1)
        return
    import torch
    import main as M
    from dlib.models.select_model import define_model
    rows = []
    for net_type, method in NETS:
        if only is not None and net_type not in only:
            continue
        argv = ["--net_type", net_type, "--method", method, "--task", "super-resolution", "--scale", str(a.scale), "--n_channels", "1",
                "--h_size", "512", "--batch_size", str(a.batch), "--G_optimizer_lr", "1e-4"]
        args = M.parse_input(argv)
        torch.manual_seed(0)
        row = {"net_type": net_type, "scale": a.scale, "batch": a.batch}
        try:
            model = define_model(args)
            model.init_train()
            model.feed_data(M.synth_batch(a.batch, a.scale, 512, model.device, 7))
            for k in range(2):
                model.optimize_parameters(0, k)
            first = model.current_log()["G_loss"]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(a.steps):
                model.optimize_parameters(0, 2 + k)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            last = model.current_log()["G_loss"]
            row.update(ms_per_step=ms, patches_per_s=a.batch / ms * 1e3, loss_after_2_steps=float(first), loss_at_end=float(last),
                       finite=bool(model.check_finite()), peak_memory_gib=torch.cuda.max_memory_allocated() / 2 ** 30)
            del model
        except Exception as e:      # a sweep row, not a gate: the tests are the gate
            row["error"] = f"{type(e).__name__}: {e}"[:300]
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
