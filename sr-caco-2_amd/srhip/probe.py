"""Live per-kernel timing for bench.py's roofline figure: HIP events recorded
on the launching stream around every launch of one op class, inside the timed
region.  Off by default (zero overhead beyond one attribute check)."""
import torch

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
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


def collect():
    """Dominant class (largest summed duration) -> roofline dict."""
    torch.cuda.synchronize()
    best = None
    for key, evs in _records.items():
        ms = sum(a.elapsed_time(b) for a, b, _ in evs)
        fl = sum(f for _, _, f in evs)
        if best is None or ms > best[1]:
            best = (key, ms, fl, len(evs))
    if best is None:
        return None
    key, ms, fl, n = best
    tf = fl / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / F32_MFMA_PEAK_TFLOPS, "traffic": None,
            "kernel": f"k_nt f32-MFMA NT GEMM M={key[1]} N={key[2]} K={key[3]}",
            "launches": n, "avg_launch_us": 1000.0 * ms / n,
            "algorithmic_gflop_per_launch": fl / n / 1e9}
