#!/usr/bin/env python3
"""Does running two half-batch GEMM chains on two streams overlap one kernel's
load/epilogue phases with the other's MFMA phase?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops

dev = "cuda"
def mk(T, N, K):
    return (torch.randn(T, K, device=dev), torch.randn(N, K, device=dev) * 0.1, torch.randn(N, device=dev),
            torch.randn(T, N, device=dev), torch.empty(T, N, device=dev))

def chain(bufs, reps):
    for _ in range(reps):
        for (A, W, b, R, out) in bufs:
            ops.gemm_nt(A, W, b, out=out, epi=2, R=R)

def run(nstreams, T, reps=10):
    shapes = [(180, 180), (360, 180), (540, 180), (180, 360)]
    sets = [[mk(T, N, K) for (N, K) in shapes] for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    torch.cuda.synchronize()
    for it in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for s, bufs in zip(streams, sets):
            s.wait_event(a)
            with torch.cuda.stream(s):
                chain(bufs, reps)
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        b.record()
        torch.cuda.synchronize()
    fl = sum(2.0 * T * N * K for (N, K) in shapes) * reps * nstreams
    ms = a.elapsed_time(b)
    print(f"streams={nstreams} T={T}: {ms:.2f} ms  {fl / ms / 1e9:.1f} TF/s", flush=True)

run(1, 32768)
run(2, 16384)
run(4, 8192)
run(2, 32768)
