"""Live per-kernel timing for bench.py's roofline figure: HIP events recorded
on the launching stream around every launch of one op class, inside the timed
region.  Off by default (zero overhead beyond one attribute check)."""
import torch

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
# bf16x3 path: every f32 product costs 6 bf16 products on v_mfma_f32_32x32x16_bf16
# (dense bf16 peak 2.5 PFLOP/s, MI355X_MICROARCH.md) -> ceiling in f32-equivalent flops
BX3_MFMA_PEAK_TFLOPS = 2500.0 / 6.0
active = None
_records = {}


def enable(kind):
    global active
    active = kind
    _records.clear()


def disable():
    global active
    active = None


class timed:
    """with probe.timed(("gemm_nt", M, N, K), flops): launch..."""

    def __init__(self, key, flops):
        self.key, self.flops = key, flops

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        _records.setdefault(self.key, []).append((self.a, self.b, self.flops))


def _kernel_name(n, k):
    """Instantiation the NT dispatcher (csrc/gemm_ntb.hip -> gemm_ntp.hip, gemm_nt.hip) picks for (N, K) at M >= 32768."""
    from . import ops
    if ops.use_bx3():
        return "k_ntp<3>" if n % 180 == 0 else None
    if n % 180 == 0 and n // 180 == 2:
        return "k_nt<2, 3, 36, false>"
    if n % 180 == 0:
        return "k_nt<1, 3, 60, false>" if k % 60 == 0 else "k_nt<1, 3, 36, false>"
    return None


def _pmc_traffic(kernel):
    """HBM bytes per launch of that kernel from the committed rocprofv3 --pmc run
    (profiles/r01_hbm_traffic_per_kernel.json; FETCH_SIZE x2 + WRITE_SIZE, separate
    passes, see tools/collect_traffic.sh).  None if not collected."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                        "profiles", "r01_hbm_traffic_per_kernel.json")
    if kernel is None or not os.path.isfile(path):
        return None
    for name, v in json.load(open(path)).items():
        if kernel in name:
            return v["hbm_bytes_per_launch"]
    return None


def collect():
    """Dominant KERNEL (largest summed duration over all the (M, N, K) classes that
    dispatch to it) -> roofline dict.  avg_launch_us is the mean over all its timed
    launches, i.e. the figure rocprofv3 --stats reports for that kernel."""
    torch.cuda.synchronize()
    from . import ops
    per_kernel = {}
    for key, evs in _records.items():
        kname = _kernel_name(key[2], key[3]) or f"nt(N={key[2]},K={key[3]})"
        ms = sum(a.elapsed_time(b) for a, b, _ in evs)
        fl = sum(f for _, _, f in evs)
        g = per_kernel.setdefault(kname, [0.0, 0.0, 0, {}])
        g[0] += ms
        g[1] += fl
        g[2] += len(evs)
        g[3][f"M={key[1]} N={key[2]} K={key[3]}"] = {"launches": len(evs), "avg_us": 1000.0 * ms / len(evs)}
    if not per_kernel:
        return None
    kname, (ms, fl, n, classes) = max(per_kernel.items(), key=lambda kv: kv[1][0])
    tf = fl / (ms * 1e-3) / 1e12
    bx = ops.use_bx3()
    peak = BX3_MFMA_PEAK_TFLOPS if bx else F32_MFMA_PEAK_TFLOPS
    what = "bf16x3-split MFMA (f32-equivalent flops; peak = bf16 dense / 6)" if bx else "f32-MFMA"
    traffic = _pmc_traffic(kname)
    out = {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
           "frac": tf / peak, "traffic": traffic,
           "kernel": f"{kname}: {what} NT GEMM", "launches": n, "avg_launch_us": 1000.0 * ms / n,
           "algorithmic_gflop_per_launch": fl / n / 1e9, "classes": classes}
    if traffic:      # the kernel sits near the ridge: its HBM side, from the PMC bytes, for reference
        out["hbm_tb_per_s_from_traffic"] = traffic / (ms / n * 1e-3) / 1e12
        out["hbm_frac_of_8tb_per_s"] = out["hbm_tb_per_s_from_traffic"] / 8.0
    return out
