#!/usr/bin/env python3
"""One network's evaluation forward (model.test()) a few times, for a profiler:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_grl -- python3 tools/mb_eval_net.py GRL GRL 8 [iters] [amp]

Same construction path as tools/eval_sweep.py (main.parse_input / define_model, synthetic 512x512 HR patches, batch 8)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch  # noqa: E402


def main():
    net_type, method, scale = sys.argv[1], sys.argv[2], int(sys.argv[3])
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    amp = len(sys.argv) > 5 and sys.argv[5] in ("1", "amp", "True")
    graph = len(sys.argv) > 6 and sys.argv[6] in ("1", "graph", "True")
    import main as M
    from dlib.models.select_model import define_model
    args = M.parse_input(["--net_type", net_type, "--method", method, "--task", "super-resolution", "--scale", str(scale),
                          "--n_channels", "1", "--h_size", "512", "--batch_size", "8", "--amp", str(amp), "--eval_graph", str(graph)])
    torch.manual_seed(0)
    model = define_model(args)
    model.init_train()
    model.feed_data(M.synth_batch(8, scale, 512, model.device, 7))
    for _ in range(2):
        model.test()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        model.test()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print(f"{net_type} x{scale} amp={amp}: {ms:.2f} ms / batch of 8, {8e3 / ms:.1f} patches/s")


if __name__ == "__main__":
    main()
