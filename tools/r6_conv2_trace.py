#!/usr/bin/env python
"""Per-workgroup timeline (instrumented build: csrc/build.sh -DUD_TRACE [-DUD_TRACE_DRAIN], UNIDEPTH_HIP_LIB=ab/libtrace.so) of the decoder's stage-2
convolutions: the residual-accumulate launch (conv2) beside the same operands through the fp16 epilogue.  Stamps per (workgroup, tile): 0 tile start,
1 first fragments read, 2 K loop done, 3 epilogue issued, 4 stores drained (UD_TRACE_DRAIN), 5 end of tile.  GPU box only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth
from unidepth_amd import UniDepthV2, ops

captured = []
orig = ops.Program.gemm
def spy(self, **kw):
    captured.append(dict(kw)); return orig(self, **kw)
ops.Program.gemm = spy
cfg = synth.load_config("vitl14")
model = UniDepthV2(cfg).load_state_dict(synth.make_synthetic_checkpoint(cfg, 125)).to("cuda").eval()
rgb = torch.randint(0, 256, (8, 3, 518, 518), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
model.infer(rgb); torch.cuda.synchronize()
ops.Program.gemm = orig
names = {v.data_ptr(): k for k, v in model._w.items() if isinstance(v, torch.Tensor) and v.is_cuda}
trace = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
ops.lib.ud_trace_set.argtypes = [C.c_void_p]
assert ops.lib.ud_trace_set(trace.data_ptr()) == 0
match = sys.argv[1] if len(sys.argv) > 1 else "dh.ups.2.0.conv2"
for kw in captured:
    name = kw.get("tag") or names.get(kw["W"].data_ptr() if isinstance(kw["W"], torch.Tensor) else kw["W"], "?")
    if match not in name:
        continue
    kw = {k: v for k, v in kw.items() if k not in ("tag", "flops", "splitk_ws", "splitk_cnt", "splitk_ws_bytes")}
    M, N = kw["M"], kw["N"]
    o16 = torch.zeros(M, N, dtype=torch.half, device="cuda")
    var = {"full": dict(kw, accumulate=1), "nopre": dict(kw, accumulate=0), "copyonly": dict(kw, accumulate=2),
           "f16": {k: v for k, v in dict(kw, out=o16, epi=ops.UD_EPI_F16, act=ops.UD_ACT_LRELU, accumulate=0).items() if k not in ("out2", "ldc2", "act2")}}
    for vn, d in var.items():
        P = ops.Program(); P.gemm(**d)
        for _ in range(3): P.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): P.run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        trace.zero_(); P.run(); torch.cuda.synchronize()
        t = trace.cpu().view(256, 8, 8).double() * 0.01
        t0 = t[:, 0, 0][t[:, 0, 0] > 0].min()
        print(f"== {name} [{vn}] {us:.1f} us/launch ({P.meta[0][0][:50]})")
        for ti in range(8):
            v = t[:, ti, 0] > 0
            if not v.any(): break
            x = t[v, ti] - t0
            seg = lambda a, b: (x[:, b] - x[:, a])
            line = (f"  tile {ti}: n={int(v.sum()):3d} start {x[:,0].mean():6.1f} (min {x[:,0].min():6.1f} max {x[:,0].max():6.1f}) | pro {seg(0,1).mean():5.2f} (max {seg(0,1).max():5.2f})"
                    f" | kloop {seg(1,2).mean():6.2f} (min {seg(1,2).min():6.2f} max {seg(1,2).max():6.2f}) | epi {seg(2,3).mean():5.2f} (max {seg(2,3).max():5.2f})")
            if (x[:, 4] > 0).any(): line += f" | drain {seg(3,4).mean():5.2f} (max {seg(3,4).max():5.2f})"
            line += f" | end {x[:,5].mean():6.1f} (min {x[:,5].min():6.1f} max {x[:,5].max():6.1f})"
            print(line)
    break
