"""Live per-kernel timing for bench.py's roofline figure: HIP events recorded
on the launching stream around every launch of the probed op classes, inside
the timed region.  Off by default (zero overhead beyond one attribute check).

Op classes ("kinds") and the kernels behind them:
  gemm_nt     Linear forward / data gradient            k_nth2 / k_ntw (gemm_ntw.hip; fp16x2 / bf16x3 planes), k_nt* exact f32
  conv_nt     3x3 conv forward / data gradient          k_nhcw / k_nhcw2<..> implicit GEMM (gemm_ntw.hip), k_ntb (gemm_ntb.hip)
  conv_tn     3x3 conv weight gradient                  k_tnb<W> / k_tnb3 conv forms (gemm_tnb.hip)
  linear_tn   Linear weight gradients (grouped launch)  k_tnb_grouped_h<W> (gemm_tnb.hip)
  wattn       window attention core, backward           k_wattn3_bwd (wattn2.hip); exact f32: k_wattn_* (wattn.hip)
  wmsa_fused  norm1 + qkv + attention + proj, forward   k_wmsa_f16h (wmsa_f16.hip)
  mlp_fused   LN -> fc1 -> GELU -> fc2 (+ its backward,  k_mlp_f16<bwd> (mlp_f16.hip)
              qkv / proj data gradients chained)
Every record carries the launch's ALGORITHMIC flops and bytes (SURVEY 8d
convention: operands read once, result written once, fp32)."""
import json
import os

import torch

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
# bf16x3 path (the fallback split): every f32 product costs 6 bf16 products on v_mfma_f32_32x32x16_bf16 (dense 16-bit peak
# 2.5 PFLOP/s, MI355X_MICROARCH.md) -> ceiling in f32-equivalent flops.  The fp16x2 path of the hot kernels costs 3 products
# (v_mfma_f32_16x16x32_f16, same dense peak): its ceiling is twice this one (bench.py: flop_frac_of_fp16x2_peak).
BX3_MFMA_PEAK_TFLOPS = 2500.0 / 6.0
HBM_PEAK_GBS = 8000.0
active = None           # None | set of kinds being timed in the current step
_records = {}
_meta = {}              # key -> (kernel name stem | None, fused-lower-bound bytes per launch)


def enable(kinds):
    global active
    active = set([kinds] if isinstance(kinds, str) else kinds)
    _records.clear()


def disable():
    global active
    active = None


def csrc_hash():
    """sha256 (16 hex digits) over the kernel sources the library is built from (csrc/*.hip, *.h, Makefile): ties a
    committed PMC profile to the build it measured."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def on(kind):
    return active is not None and kind in active


class timed:
    """with probe.timed(("gemm_nt", M, N, K), flops, bytes): launch..."""

    def __init__(self, key, flops, nbytes=0.0, kernel=None, lb_bytes=None):
        """kernel: name stem of THE kernel behind this launch (rocprofv3 / PMC tables are keyed by kernel name);
        lb_bytes: SURVEY 8d's fused lower bound for the launch -- only the op's inputs, outputs and weights touch HBM,
        no saved intermediate -- beside nbytes, the operand bytes this launch is designed to move."""
        self.key, self.flops, self.nbytes = key, flops, nbytes
        _meta[key] = (kernel, nbytes if lb_bytes is None else lb_bytes)

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        _records.setdefault(self.key, []).append((self.a, self.b, self.flops, self.nbytes))


_KERNEL_OF_KIND = {
    "gemm_nt": ("k_nth|k_ntw|k_ntp", "NT GEMM (Linear forward / data gradient)"),
    "conv_nt": ("k_nhcw|k_ntcw|k_ntb", "implicit-GEMM 3x3 conv (forward / data gradient)"),
    "conv_tn": ("k_tnb", "3x3 conv weight gradient"),
    "linear_tn": ("k_tnb_grouped", "grouped Linear weight gradients"),
    "wattn": ("k_wattn", "window attention core"),
    "mlp_fused": ("k_mlp", "fused LayerNorm -> fc1 -> GELU -> fc2 -> residual (and its backward, with the proj data "
                           "gradient chained)"),
    "wmsa_fused": ("k_wmsa", "fused LayerNorm -> qkv -> window attention -> proj -> residual (forward)"),
    # the evaluation sweep's other classes (tools/eval_sweep.py): exact-f32 arithmetic, priced against the f32 peak
    "gemm_nt_f32": ("k_nt<", "NT GEMM, exact f32 (v_mfma_f32_32x32x2_f32)"),
    "conv_nt_f32": ("k_nt<", "implicit-GEMM 3x3 conv, exact f32"),
    "gemm_batched": ("k_nt<", "batched NT GEMM, exact f32: every (sample, head) product of a layer in one launch"),
    "nlsa": ("k_nlsa_attention", "NLSN chunk attention on the exact-f32 matrix core"),
    "grl_attn": ("k_cosine_window_attention", "GRL cosine window / anchor-stripe attention (vector ALU, f32)"),
    # fp16 STORAGE (--amp evaluation, conv_h16.hip): one fp16 product per multiply, fp16 maps: priced against the dense
    # 16-bit matrix peak and 2-byte elements
    "conv_h16": ("k_conv3x3_h16|k_conv1x1_h16|k_srcnn_h16", "3x3 / 1x1 conv on fp16 storage (one v_mfma_f32_16x16x32_f16 product, "
                                                              "f32 accumulate), SRCNN's three layers in one launch"),
    # OmniSR's own kernels (omni_ops.hip): depthwise convs, window / grid attention, channel attention -- vector ALU, f32
    "omni_ops": ("k_dwconv3x3|k_group_attention|k_channel_attention", "OmniSR depthwise 3x3 convs, 64-token window / grid attention, "
                                                                    "channel attention (vector ALU, f32)"),
}
_F32_KINDS = ("gemm_nt_f32", "conv_nt_f32", "gemm_batched", "nlsa", "grl_attn", "omni_ops")
ALL_KINDS = tuple(_KERNEL_OF_KIND)


def _pmc_traffic(kernel_stem):
    """Average HBM bytes per launch of the kernels whose name contains ``kernel_stem``, from the
    committed rocprofv3 --pmc passes (profiles/r0N_hbm_traffic_per_kernel.json; FETCH_SIZE x2 +
    WRITE_SIZE, separate passes, tools/collect_traffic.sh).  None if not collected."""
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    want = os.environ.get("SRHIP_TRAFFIC_JSON")
    cands = [want] if want else [os.path.join(root, "profiles", f"r0{r}_hbm_traffic_per_kernel.json")
                                 for r in (9, 8, 7, 6, 5, 4, 3, 2, 1)]
    for path in cands:
        if path and os.path.isfile(path):
            table = json.load(open(path))
            # counters of ANOTHER build of the kernels say nothing about this run: a profile is used only if it was
            # collected from the kernel sources this library was built from (tools/parse_traffic.py records their hash)
            if table.get("_meta", {}).get("csrc_sha16") != csrc_hash():
                continue
            for st in kernel_stem.split("|"):        # alternatives in order of preference: the first one that ran
                tot = n = 0.0
                for name, v in table.items():
                    if name != "_meta" and st in name:
                        tot += v["hbm_bytes_per_launch"] * v["launches"]
                        n += v["launches"]
                if n:
                    return tot / n, os.path.basename(path)
    return None, None


def _pmc_mfma_busy(kernel_stem):
    """MFMA-busy share (SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)) of the kernels whose name contains
    kernel_stem, from the committed, hash-matched profiles/r0N_mfma_busy_per_kernel_<workload>_b8.json (tools/step_mfma_pmc.sh).
    None if not collected for THIS build of the kernels."""
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    want = os.environ.get("SRHIP_TRAFFIC_JSON")
    if not want:
        return None, None
    path = want.replace("hbm_traffic_per_kernel", "mfma_busy_per_kernel")
    if not os.path.isfile(path):
        return None, None
    table = json.load(open(path))
    if table.get("_meta", {}).get("csrc_sha16") != csrc_hash():
        return None, None
    tot = n = 0.0
    for name, v in table.items():
        if name != "_meta" and kernel_stem in name and "mfma_busy_frac_at_2.4GHz" in v:
            tot += v["mfma_busy_frac_at_2.4GHz"] * v["launches"]
            n += v["launches"]
    return (tot / n, os.path.basename(path)) if n else (None, None)


def collect():
    """Dominant KERNEL -- the (op class, instantiation, shape) key with the largest summed launch time -> roofline dict
    (round 6; the class sum rides along under `class_*`).  avg_launch_us is the mean over its timed launches, i.e. the figure
    rocprofv3 --stats reports for that kernel."""
    torch.cuda.synchronize()
    from . import ops
    # what an event pair measures with NOTHING between its records: the part of avg_launch_us that is not the kernel
    pairs = []
    for _ in range(64):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    gaps = sorted(a.elapsed_time(b) for a, b in pairs)
    pair_us = 1000.0 * gaps[len(gaps) // 2]
    per_kind = {}
    for key, evs in _records.items():
        ms = sum(a.elapsed_time(b) for a, b, _, _ in evs)
        g = per_kind.setdefault(key[0], [0.0, 0.0, 0.0, 0, {}])
        g[0] += ms
        g[1] += sum(e[2] for e in evs)
        g[2] += sum(e[3] for e in evs)
        g[3] += len(evs)
        g[4][" ".join(str(v) for v in key[1:])] = {"launches": len(evs), "avg_us": 1000.0 * ms / len(evs)}
    if not per_kind:
        return None
    # the dominant kernel: one key = one kernel instantiation at one shape
    per_key = {key: (sum(a.elapsed_time(b) for a, b, _, _ in evs), sum(e[2] for e in evs), sum(e[3] for e in evs), len(evs))
               for key, evs in _records.items()}
    dom_key, (ms, fl, by, n) = max(per_key.items(), key=lambda kv: kv[1][0])
    kind = dom_key[0]
    class_ms, class_fl, class_by, class_n, classes = per_kind[kind]
    stem, what = _KERNEL_OF_KIND.get(kind, (kind, kind))
    kname, lb_per_launch = _meta.get(dom_key, (None, by / n))
    if kname:
        stem = kname
    tf = fl / (ms * 1e-3) / 1e12
    gbs = by / (ms * 1e-3) / 1e9
    bx = ops.use_bx3() and kind not in _F32_KINDS
    peak = BX3_MFMA_PEAK_TFLOPS if bx else F32_MFMA_PEAK_TFLOPS
    arith = "bf16x3-split MFMA (f32-equivalent flops; peak = bf16 dense / 6)" if bx else "f32 MFMA"
    if bx and kind == "conv_nt" and ops.F16X2_CONV:    # 64 .. 256-channel convs: three products on two fp16 planes
        peak = 2500.0 / 3.0
        arith = ("fp16x2-split MFMA: two fp16 planes per operand, power-of-two scales per weight output channel and per "
                 "activation halo tile, three products, f32 accumulate (f32-equivalent flops; peak = fp16 dense / 3; frac "
                 "against the six-product bf16x3 ceiling 416.7 in frac_of_bf16x3_peak)")
    if bx and (kind in ("mlp_fused", "wmsa_fused") or (kind == "wattn" and ops.WATTN_F16)):
        peak = 2500.0 / 3.0
        arith = ("fp16x2-split MFMA: two fp16 planes per operand under power-of-two block exponents, three products, f32 "
                 "accumulate (f32-equivalent flops; peak = fp16 dense / 3)")
    if kind == "conv_h16":
        peak = 2500.0
        arith = "fp16 MFMA, one product per multiply on fp16 storage (peak = fp16 dense)"
    if bx and kind == "gemm_nt" and ops.F16X2:      # three products on two fp16 planes: the ceiling of THIS algorithm is twice as high
        peak = 2500.0 / 3.0
        arith = ("fp16x2-split MFMA: two fp16 planes per operand under per-row power-of-two scales, three products, f32 "
                 "accumulate (f32-equivalent flops; peak = fp16 dense / 3; frac against the six-product bf16x3 ceiling "
                 "416.7 in frac_of_bf16x3_peak)")
    traffic, src = _pmc_traffic(stem)
    # the roofline that bounds this class: time >= max(algorithmic bytes / HBM peak, algorithmic flops / MFMA peak)
    hbm_bound = by / (HBM_PEAK_GBS * 1e9) > fl / (peak * 1e12)
    out = {"bound": "hbm" if hbm_bound else "mfma", "achieved": gbs if hbm_bound else tf,
           "peak": HBM_PEAK_GBS if hbm_bound else peak, "unit": "GB/s" if hbm_bound else "TFLOP/s",
           "frac": gbs / HBM_PEAK_GBS if hbm_bound else tf / peak,
           "mfma_side": {"achieved_tflops": tf, "peak_tflops": peak, "frac": tf / peak},
           "traffic": traffic, "traffic_source": src,
           "kernel": f"{stem.split('|')[0]}*: {what}; {arith}", "kernel_key": " ".join(str(v) for v in dom_key),
           "launches": n, "avg_launch_us": 1000.0 * ms / n,
           # achieved / frac above are priced on avg_launch_us as measured (conservative); an empty event pair on this
           # stream reads event_pair_us, so the kernel itself takes about avg_launch_us - event_pair_us -- the figure to
           # hold against rocprofv3's average duration in profiles/
           "event_pair_us": pair_us, "avg_launch_us_minus_event_pair": 1000.0 * ms / n - pair_us,
           "algorithmic_gflop_per_launch": fl / n / 1e9,
           "algorithmic_mbytes_per_launch": by / n / 1e6,
           "hbm_side": {"achieved_gb_per_s_algorithmic": gbs, "frac_of_8tb_per_s": gbs / HBM_PEAK_GBS},
           # two accountings of the same launch: `frac` prices the operand bytes the kernel is DESIGNED to
           # move (saved activations of a training step included); this one SURVEY 8d's fused lower bound (inputs, outputs,
           # weights only) -- against the same binding roof
           "fused_lower_bound_mbytes_per_launch": lb_per_launch / 1e6,
           "frac_vs_fused_lower_bound": max(lb_per_launch / (HBM_PEAK_GBS * 1e9), fl / n / (peak * 1e12)) / (ms / n * 1e-3),
           "class": {"kind": kind, "launches": class_n, "summed_ms": class_ms,
                     "frac": max(class_by / (HBM_PEAK_GBS * 1e9), class_fl / (peak * 1e12)) / (class_ms * 1e-3)},
           "probed_ms": {k: v[0] for k, v in per_kind.items()},
           "share_of_probed_time": {k: v[0] / sum(x[0] for x in per_kind.values()) for k, v in per_kind.items()},
           "classes": classes}
    if bx and ((kind == "gemm_nt" and ops.F16X2) or (kind == "conv_nt" and ops.F16X2_CONV)):
        out["frac_of_bf16x3_peak"] = tf / BX3_MFMA_PEAK_TFLOPS
    if traffic:
        out["hbm_side"]["measured_gb_per_s_from_pmc_traffic"] = traffic / (ms / n * 1e-3) / 1e9
        out["traffic_over_charged_bytes"] = traffic / (by / n)
    busy, busy_src = _pmc_mfma_busy(stem.split("|")[0])
    out["mfma_busy"] = busy                      # None unless profiles/ holds a pass over THIS build of the kernels
    out["mfma_busy_source"] = busy_src
    return out
