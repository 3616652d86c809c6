#!/usr/bin/env python3
"""Does putting the two blocks of a CU in different phases pay?  Experiments build: SRHIP_MLP_STAGGER (10-ns units) delays
either odd blocks (mode 0) or the block that arrives SECOND on its CU (mode 1, per-CU arrival counters); cold operands
(a 512-MiB fill between launches evicts the Infinity Cache).  Also prints which blocks share a CU."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops

C, hid = 180, 360
dev = "cuda"
evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
cuc = torch.zeros(4096, dtype=torch.int32, device=dev)
f_cu = ops.lib.srhip_mlp_cu_counters; f_cu.argtypes = [ctypes.c_void_p]; f_cu(cuc.data_ptr())
f_dbg = ops.lib.srhip_mlp_debug_buffer; f_dbg.argtypes = [ctypes.c_void_p]

def setup(M):
    x = torch.randn(M, C, device=dev); dy = torch.randn(M, C, device=dev)
    w1 = torch.randn(hid, C, device=dev) * 0.1; w2 = torch.randn(C, hid, device=dev) * 0.1
    b1 = torch.randn(hid, device=dev); b2 = torch.randn(C, device=dev)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    P = {k: ops.Bx3(*s, dev) for k, s in dict(w1=(hid, C), w2=(C, hid), w1T=(C, hid), w2T=(hid, C)).items()}
    b1f = torch.empty(hid, device=dev)
    K0 = 3 * C
    X0 = torch.randn(M, K0, device=dev); x0 = torch.randn(M, C, device=dev); res0 = torch.randn(M, C, device=dev)
    st0 = torch.empty(M, 2, device=dev); ops.layernorm_fwd(x0, st0)
    w0 = torch.randn(K0, C, device=dev) * 0.1; w3 = torch.randn(C, C, device=dev) * 0.1
    P["w0T"] = ops.Bx3(C, K0, dev); P["w3T"] = ops.Bx3(C, C, dev)
    tb = ops.PrepTable()
    tb.linear(w1, P["w1"], gamma=gamma); tb.linear(w1, P["w1T"], gamma=gamma, transpose=True)
    tb.linear(w2, P["w2"]); tb.linear(w2, P["w2T"], transpose=True)
    tb.fold_bias(w1, b1, beta, b1f)
    tb.linear(w0, P["w0T"], transpose=True); tb.linear(w3, P["w3T"], transpose=True)
    tb.build(dev).run()
    st = torch.empty(M, 2, device=dev); ops.layernorm_fwd(x, st)
    h = torch.empty(M, hid, device=dev); out = torch.empty(M, C, device=dev); sto = torch.empty(M, 2, device=dev)
    dh = torch.empty(M, hid, device=dev); gh = torch.empty(M, hid, device=dev); dx = torch.empty(M, C, device=dev)
    out3 = torch.empty(M, C, device=dev)
    def fwd(): ops.mlp_fwd_f16(x, st, P["w1"], b1f, P["w2"], b2, out, h=h, stats_out=sto)
    def bwd(): ops.mlp_bwd_f16(dy, P["w2T"], P["w1T"], h, dh, gh, x, st, dx)
    def bwd_full(): ops.mlp_bwd_f16(dy, P["w2T"], P["w1T"], h, dh, gh, x, st, dx, chain=(P["w3T"], out3, None),
                                    front=(X0, P["w0T"], x0, st0, res0))
    return dict(fwd=fwd, bwd=bwd, bwd_full=bwd_full), (out, h, dx, dh, gh, out3)

def timeit(f, n=9):
    t = []
    for _ in range(n):
        evict.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1) * 1e3)
    t.sort()
    return t[n // 2], t[0]

for M in (32768, 16384):
    fns, outs = setup(M)
    # co-residency map
    nblk = M // 64
    dbg = torch.zeros(nblk, 4, 32, dtype=torch.int64, device=dev)
    os.environ["SRHIP_MLP_STAGGER"] = "0"; os.environ["SRHIP_MLP_STAGGER_MODE"] = "1"
    f_dbg(dbg.data_ptr()); fns["fwd"](); torch.cuda.synchronize(); f_dbg(None)
    cu = dbg[:, 0, 24].cpu().tolist(); slot = dbg[:, 0, 25].cpu().tolist()
    by = {}
    for b, c_ in enumerate(cu): by.setdefault(c_, []).append(b)
    sizes = {}
    for v in by.values(): sizes[len(v)] = sizes.get(len(v), 0) + 1
    print(f"M={M}: {nblk} blocks on {len(by)} CUs; blocks per CU histogram {sizes}; examples {list(by.items())[:6]}")
    diffs = {}
    for v in by.values():
        if len(v) == 2: diffs[v[1] - v[0]] = diffs.get(v[1] - v[0], 0) + 1
    print("   block-index distance of CU partners:", sorted(diffs.items(), key=lambda kv: -kv[1])[:6])
    ref = {}
    for mode in (1, 0):
        for stg in (0, 200, 400, 600, 900, 1200, 1800):
            if mode == 0 and stg == 0: continue
            os.environ["SRHIP_MLP_STAGGER"] = str(stg); os.environ["SRHIP_MLP_STAGGER_MODE"] = str(mode)
            row = []
            for name, f in fns.items():
                for _ in range(2): f()
                med, mn = timeit(f)
                row.append(f"{name} {med:6.1f} (min {mn:6.1f})")
            print(f"  M={M} mode {mode} stagger {stg / 100:5.1f} us: " + "  ".join(row), flush=True)
