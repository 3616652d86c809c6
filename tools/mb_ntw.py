#!/usr/bin/env python3
"""A/B of the NT GEMM with W fragments straight from global memory (gemm_ntw.hip, SRHIP_NTW=1, the default) against
the LDS-staged one (gemm_ntp.hip, SRHIP_NTW=0) on the Linears of a Swin block at T = 32768 rows, with their real
prologues / epilogues, back to back (operands L2 / Infinity-Cache warm) and cycling through 12 buffer sets (every
launch sees cold inputs, as inside the training step); then the timing ablations of k_ntw (SRHIP_NTW_DBG bits: results
wrong on purpose).  Same box, one process per arm.

    python tools/mb_ntw.py [dbg bits ...]      e.g.  python tools/mb_ntw.py 1 2 4 8 6"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))

SHAPES = [  # name, N, K, a_mode, epi, stats_out, aux
    ("qkv fwd   LN-pro", 540, 180, 1, 0, False, False),
    ("proj fwd  res+stats", 180, 180, 0, 2, True, False),
    ("fc1 fwd   LN-pro", 360, 180, 1, 0, False, False),
    ("fc2 dgrad gelu'+aux", 360, 180, 0, 3, False, True),
    ("proj dgrad scale", 180, 180, 0, 2, False, False),
    ("fc2 fwd gelu-pro+res", 180, 360, 2, 2, True, False),
]


def arm():
    import torch
    import torch.nn.functional as F
    from srhip import ops
    T, dev, NSET = 32768, "cuda", 12
    torch.manual_seed(0)
    for name, N, K, a_mode, epi, want_stats, want_aux in SHAPES:
        W = torch.randn(N, K, device=dev) * 0.1
        b = torch.randn(N, device=dev)
        Wb = ops.split_bf16x3(W)
        rs = torch.tensor([1.0, 0.0, 1.25, 2.0, 1.0, 1.0, 0.5, 1.0], device=dev)
        sets = []
        pad = lambda n: (n + 31) // 32 * 32 if os.environ.get("MB_PAD") else n      # MB_PAD=1: 128-byte aligned row pitches
        for _ in range(NSET):
            A = torch.randn(T, pad(K), device=dev)[:, :K]
            st = torch.stack([A.mean(1), 1 / torch.sqrt(A.var(1, unbiased=False) + 1e-5)], 1).contiguous()
            sets.append(dict(A=A, st=st, R=torch.randn(T, pad(N), device=dev)[:, :N], out=torch.empty(T, pad(N), device=dev)[:, :N],
                             so=torch.empty(T, 2, device=dev), aux=torch.empty(T, pad(N), device=dev)[:, :N]))

        def run(s):
            ops.gemm_nt(s["A"], Wb, b if epi != 3 else None, out=s["out"], a_mode=a_mode,
                        ln_stats=s["st"] if a_mode == 1 else None, epi=epi, R=s["R"] if epi >= 2 else None,
                        rowscale=rs if epi >= 2 else None, rows_per_scale=T // 8,
                        aux=s["aux"] if want_aux else None, stats_out=s["so"] if want_stats else None)

        def timeit(cold, n=36):
            for i in range(NSET):
                run(sets[i if cold else 0])
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(n):
                run(sets[i % NSET if cold else 0])
            e.record()
            torch.cuda.synchronize()
            return a.elapsed_time(e) / n * 1e3
        warm, cold = timeit(False), timeit(True)
        # correctness on a row sample against float64
        s = sets[0]
        run(s)
        idx = torch.arange(0, T, 97, device=dev)
        Ad = s["A"][idx].double()
        if a_mode == 1:
            Ad = (Ad - s["st"][idx, :1].double()) * s["st"][idx, 1:].double()
        elif a_mode == 2:
            Ad = F.gelu(Ad)
        y = F.linear(Ad, W.double(), None if epi == 3 else b.double())
        sc = rs.double()[idx // (T // 8)][:, None]
        if epi == 2:
            y = s["R"][idx].double() + sc * y
        elif epi == 3:
            Rg = s["R"][idx].double().requires_grad_(True)
            F.gelu(Rg).sum().backward()
            y = sc * y * Rg.grad
        err = ((s["out"][idx].double() - y).abs().max() / y.abs().max()).item()
        extra = ""
        if want_stats:
            full = s["out"].double()
            m, r = full.mean(1), 1 / torch.sqrt(full.var(1, unbiased=False) + 1e-5)
            extra = f" stats err {max(((s['so'][:, 0].double() - m).abs().max() / m.abs().max()).item(), ((s['so'][:, 1].double() - r).abs().max() / r.abs().max()).item()):.1e}"
        if want_aux:
            extra += f" aux err {((s['aux'][idx].double() - F.gelu(s['R'][idx].double())).abs().max()).item():.1e}"
        fl = 2.0 * T * N * K
        print(f"  {name:22s} N={N:3d}: warm {warm:6.1f} us ({fl / warm * 1e-6:6.1f} TF/s)  cold {cold:6.1f} us "
              f"({fl / cold * 1e-6:6.1f} TF/s)  rel err {err:.1e}{extra}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "arm":
        arm()
    else:
        arms = [dict(SRHIP_NTW="0"), dict(SRHIP_NTW="1"), dict(SRHIP_NTW="1", MB_PAD="1")] + \
            [dict(SRHIP_NTW="1", SRHIP_NTW_DBG=b) for b in sys.argv[1:]]
        for env in arms:
            print(env, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "arm"], env=dict(os.environ, **env), check=False)
