#!/usr/bin/env python
"""ud_dwconv7_nhwc_f32 at the four ConvNeXt-L stage shapes of BASELINE configs[3] (640x480, bs 16): us per launch, TB/s of compulsory
traffic (read + write of the map), TFLOP/s.  For A/B-ing library builds (UNIDEPTH_HIP_LIB=, e.g. ablations -DUD_DW_NOCOMPUTE / NOSTORE / NODMA)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops
B = 16
out = []
for (H, W, C) in ((120, 160, 192), (60, 80, 384), (30, 40, 768), (15, 20, 1536)):
    x = torch.randn(B, H, W, C, device="cuda"); y = torch.zeros_like(x)
    w = torch.randn(49, C, device="cuda"); b = torch.randn(C, device="cuda")
    d = ops.mk(ops.UdDwConv7, x=x, w=w, bias=b, y=y, B=B, H=H, W=W, C=C, ldx=C, ldy=C)
    run = lambda: ops.check(ops.lib.ud_dwconv7_nhwc_f32(ctypes.byref(d), ops.cur_stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    n = B * H * W * C
    out.append(f"{H}x{W}x{C}: {us:7.1f} us  {8.0 * n / us / 1e6:5.2f} TB/s  {98.0 * n / us / 1e6:5.1f} TF")
print(os.environ.get("UNIDEPTH_HIP_LIB", "default"), " | ".join(out))
