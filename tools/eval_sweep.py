#!/usr/bin/env python3
"""Inference sweep over the networks libsrhip runs (BASELINE.json config 5 is the reference's 16-method
x {2,4,8} evaluation sweep, eval_all.sh): patches/s of ``model.test()`` on synthetic 512x512 HR patches,
fp32-accurate and with ``--amp True`` (reduced-precision kernels, where the network takes them), one GPU.

    python tools/eval_sweep.py [--batch 8] [--iters 20] [--nets MemNet,VDSR] [--out profiles/r02_eval_sweep.json]

(--nets: only these; with --out naming an existing file their rows replace the old ones, the others stay.)

Every network is built through the same ``main.parse_input`` / ``define_model`` path as ``main.py`` /
``eval.py``; the interpolated input the VDSR / DRRN / SRCNN family expects is produced before the timed
region (the reference builds it in the dataset, dataset_dpsr.py:700-710)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

NETS = [("swinir", "SWINIR"), ("EDSR_LIIF", "EDSR_LIIF"), ("VDSR", "VDSR"), ("DRRN", "DRRN"), ("SRCNN", "SRCNN"),
        ("MSLapSRN", "MSLAPSR"), ("MemNet", "MemNet"), ("DBPN", "DBPN"), ("SRFBN", "SRFBN"), ("ProSR", "PROSR"), ("ENLCN", "ENLCN"), ("NLSN", "NLSN"), ("DFCAN", "DFCAN"), ("ACT", "ACT"), ("OmniSR", "OmniSR"), ("GRL", "GRL")]


def roofline_of(model, forward_ms, amp_used):
    """One more forward with HIP events around every launch of the op classes srhip.probe knows (ALGORITHMIC flops / bytes per
    launch): the class with the largest summed time, its achieved rate against the ceiling of its arithmetic, and how much of
    the forward it is.  The events serialise nothing but add ~10 us per launch: shares are of the UNPROBED forward time."""
    from srhip import probe
    probe.enable(probe.ALL_KINDS)
    prev = os.environ.get("SRHIP_EVAL_GRAPH")
    os.environ["SRHIP_EVAL_GRAPH"] = "0"          # the probed forward runs eagerly (a graph replay makes no Python-side launches)
    try:
        model.test()
    finally:
        if prev is None:
            os.environ.pop("SRHIP_EVAL_GRAPH", None)
        else:
            os.environ["SRHIP_EVAL_GRAPH"] = prev
    r = probe.collect()
    probe.disable()
    if r is None:
        return {"kernel": None, "note": "no launch of a probed op class (srhip.probe) in this forward"}
    tot = sum(r["probed_ms"].values())
    dom = max(r["probed_ms"].values())
    out = {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_us",
                              "algorithmic_gflop_per_launch", "algorithmic_mbytes_per_launch")}
    if amp_used:      # the single-product kernels: the ceiling of THAT arithmetic is the dense 16-bit peak
        out["note_amp"] = "reduced-precision forward: one 16-bit product per multiply; frac is against the three-product ceiling"
    out["mfma_frac"], out["hbm_frac"] = r["mfma_side"]["frac"], r["hbm_side"]["frac_of_8tb_per_s"]
    out["share_of_forward"] = min(1.0, dom / forward_ms)
    out["probed_share_of_forward"] = min(1.0, tot / forward_ms)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--nets", default=None)
    ap.add_argument("--graph", action="store_true", help="evaluation forwards replayed from a hipGraph (--eval_graph True)")
    a = ap.parse_args()
    only = set(a.nets.split(",")) if a.nets else None
    import main as M
    from dlib.models.select_model import define_model
    rows = []
    for net_type, method in NETS:
        if only is not None and net_type not in only:
            continue
        for scale in (2, 4, 8):
            for amp in (False, True):
                argv = ["--net_type", net_type, "--method", method, "--task", "super-resolution", "--scale", str(scale),
                        "--n_channels", "1", "--h_size", "512", "--batch_size", str(a.batch), "--amp", str(amp),
                        "--eval_graph", str(bool(a.graph))]
                args = M.parse_input(argv)
                torch.manual_seed(0)
                model = define_model(args)
                model.init_train()
                batch = M.synth_batch(a.batch, scale, 512, model.device, 7)
                model.feed_data(batch)
                if net_type == "MemNet":
                    # its BatchNorms need running statistics: with the initial (0, 1) every forward grows by orders of
                    # magnitude per memory block (and leaves fp16's range under --amp).  A few training-mode forwards on a
                    # small crop, as tests/test_gpu_amp.py does
                    model.netG.train()
                    with torch.no_grad():
                        crop = model.L[:2, :, :32, :32].contiguous()
                        for _ in range(30):
                            model.netG(crop)
                    model.netG.eval()
                for _ in range(3):
                    model.test()
                torch.cuda.synchronize()
                # median over groups of iterations: a stray host stall (these forwards are 1-2 ms of GPU work) inside ONE
                # timed span once produced a 6x outlier for a whole configuration
                groups = []
                for _ in range(5):
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record()
                    for _ in range(a.iters):
                        model.test()
                    t1.record()
                    torch.cuda.synchronize()
                    groups.append(t0.elapsed_time(t1) / a.iters)
                ms = sorted(groups)[2]
                out = model.E
                assert tuple(out.shape[-2:]) == (512, 512) and torch.isfinite(out).all()
                amp_used = bool(amp and getattr(model.netG, "amp", False) and getattr(model.netG, "amp_takes_effect", True))
                rows.append({"net_type": net_type, "scale": scale, "amp_flag": amp, "reduced_precision_kernels": amp_used,
                             "batch": a.batch, "ms_per_batch": ms, "patches_per_s": a.batch / ms * 1e3})
                rows[-1]["roofline"] = roofline_of(model, ms, amp_used)
                if a.graph:
                    rows[-1]["eval_graph"] = True
                print(json.dumps(rows[-1]), flush=True)
                del model
                torch.cuda.empty_cache()
    if a.out:
        if only is not None and os.path.isfile(a.out):
            old = json.load(open(a.out))["rows"]
            rows = [r for r in old if r["net_type"] not in only or bool(r.get("eval_graph")) != bool(a.graph)] + rows
        with open(a.out, "w") as f:
            json.dump({"what": "model.test() on synthetic 512x512 HR patches, one MI355X; registry default options per network "
                               "(SwinIR: 6 x 6 blocks, embed 180); amp_flag = --amp True, reduced_precision_kernels = whether the "
                               "network takes the single-product bf16 kernels", "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
