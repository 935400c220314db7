"""Where the one-launch camera head (ud_camera_head_f32) spends its time: per-phase stamps of workgroup 0 and of the last workgroup.
Needs a library built with the stamps compiled in:
    UD_OUT=$PWD/ab/libcamtrace.so UD_BUILD_DIR=build_camtrace unidepth_amd/csrc/build.sh -DUD_CAM_TRACE
    UNIDEPTH_HIP_LIB=$PWD/ab/libcamtrace.so python tools/trace_camera_head.py [B] [workgroups]
Prints, per phase: work (x loads + LayerNorm + FMAs + stores issued), store drain, barrier wait, in microseconds (100 MHz counter)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from unidepth_amd import ops  # noqa: E402
import test_kernels_gpu as T  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ph, bufs, scale, _, keep = T._camera_head_case(ops, B, 512, 1024, 8)
ws = torch.zeros(16 + 2 * 24 * 4 * 2, dtype=torch.int32, device="cuda")
desc = ops.camera_head_desc(ph, 4, 8, 512, scale, 1e-5, ws, G)
for _ in range(5):
    ops.camera_head(desc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.camera_head(desc)
e1.record()
torch.cuda.synchronize()
print(f"B={B} workgroups={G or 128}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch; barrier flag {ws[2].item()}")
tr = ws[16:].view(torch.int64).view(2, 24, 4).cpu()
names = ["adapter0", "adapter1", "adapter2", "adapter3", "project.fc1", "project.fc2"] + [f"agg{b}.{n}" for b in (1, 2) for n in ("qkv", "attention", "out", "fc1", "fc2")] + ["out.fc1", "out.fc2"]
for who, t in (("workgroup 0", tr[0]), ("last workgroup", tr[1])):
    print(who, "  phase: work / store drain / barrier (us);  total", (t[len(ph) - 1, 3] - t[0, 0]).item() / 100.0)
    for i in range(len(ph)):
        a, b, c, d = [x.item() for x in t[i]]
        print(f"  {names[i]:14s} {(b - a) / 100.0:6.2f} {(c - b) / 100.0:6.2f} {(d - c) / 100.0:6.2f}")
