#!/usr/bin/env python3
"""Roofline of the feature up-projection MLP at the render.sh size: 1600x1200
pixels, 32 -> 128 -> 512, fp32 results.  Since r03 the default forward (gg_mlp_fwd_fast: fp16 two-piece operands on
the 16x-rate matrix instruction) needs a quarter of the matrix cycles of the fp32 kernel and is bound by its 3.9 GB of
OUTPUT: `roofline` is the HBM one (bytes read + written over time against 8 TB/s; a plain fill of the same size runs
at 6.8 TB/s on this chip: DESIGN 3.7 has the ablations), `mfma_fp32_equivalent` what the old bound would have said; `--exact` times the
exact-order fp32 kernel (gg_mlp_fwd).  Times with the in-library hipEvents, torch's
Linear/ReLU/Linear (the reference's implementation, on the same GPU) with torch events, and the CPU
oracle on a bounded sample.  Prints one JSON line."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
import torch
from gaussiangrasper_amd import _lib
from gaussiangrasper_amd.mlp import MLP

from gaussiangrasper_amd import mlp as mlp_mod
PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA, dense
HBM_PEAK_GBS = 8000.0
mlp_mod.EXACT_ORDER = "--exact" in sys.argv
dev = "cuda:0"
if "--lib" in sys.argv:                      # A/B: another build of the library
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
lib = _lib.load(build_if_missing=False) if "--lib" in sys.argv else _lib.load()
h, w, cin, cout = 1200, 1600, 32, 512
if "--config5" in sys.argv:                  # BASELINE config 5's fea_up: 128 -> 128 -> 512 at 1920x1080
    h, w, cin = 1080, 1920, 128
rows = h * w
flops = 2.0 * rows * (cin * 128 + 128 * cout)
torch.manual_seed(0)
m = MLP(cin, cout, [128]).to(dev)
img = torch.randn(h, w, cin, device=dev)
with torch.no_grad():
    for _ in range(2):
        y = m(img)
    torch.cuda.synchronize()
    lib.gg_prof_reset(); lib.gg_prof_enable(1)
    for _ in range(5):
        y = m(img)
    torch.cuda.synchronize(); lib.gg_prof_enable(0)
    n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
    lib.gg_prof_get(8, ctypes.byref(n), ctypes.byref(ms))
    ours_ms = ms.value / n.value
    for _ in range(2):
        ref = m.layers(img)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ref = m.layers(img)
    e1.record(); torch.cuda.synchronize()
    torch_ms = e0.elapsed_time(e1) / 5
    err = float((y - ref).abs().max() / ref.abs().max())
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
O.build()
sample = 200_000
t0 = time.perf_counter()
O.mlp_fwd(img.reshape(-1, cin)[:sample].cpu().numpy(), *[p.detach().cpu().numpy() for p in
          (m.layers[0].weight, m.layers[0].bias, m.layers[2].weight, m.layers[2].bias)])
cpu_s = time.perf_counter() - t0
print(json.dumps({
    "kernel": f"mlp_fwd_kernel<{cin}>" if mlp_mod.EXACT_ORDER else f"mlp_fwd_f16_kernel<{cin}>",
    "workload": f"{w}x{h} pixels, {cin}->128->512 fp32, {flops / 1e9:.1f} GFLOP, "
    f"{rows * cout * 4 / 1e9:.2f} GB written",
    "ms": ours_ms,
    "roofline": ({"bound": "mfma", "achieved": flops / ours_ms / 1e9, "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": flops / ours_ms / 1e9 / PEAK_TFLOPS} if mlp_mod.EXACT_ORDER else
                 {"bound": "hbm", "achieved": rows * (cin + cout) * 4 / ours_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": rows * (cin + cout) * 4 / ours_ms / 1e6 / HBM_PEAK_GBS}),
    "mfma_fp32_equivalent": {"achieved": flops / ours_ms / 1e9, "peak": PEAK_TFLOPS, "unit": "TFLOP/s"},
    "torch_linear_relu_linear_ms": torch_ms, "max_rel_diff_vs_torch": err,
    "cpu_baseline": {"value": sample / cpu_s, "unit": "pixels/s", "cores": O.num_threads(), "kind": "port",
                     "sample": f"{sample} pixels"}, "gpu_pixels_per_s": rows / (ours_ms * 1e-3)}))
