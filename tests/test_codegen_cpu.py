"""Build-time check of a hand-counted wait (csrc/gemm_pp.hip): the ping-pong GEMM's prologue issues the operand DMA of K-tile 0
(A: 3 instructions, W(0): 4, W(1): 4), then the 24 loads of the old fp32 values, and waits with `s_waitcnt vmcnt(28)` = "A(0) and W(0) have
landed; W(1) and the old values may still be in flight".  That is only right if the compiler keeps exactly this order and count -- the old-value
loads are plain C++ loads it may legally move.  The generated gfx950 assembly is checked here (hipcc cross-compiles without a GPU, ~3 s):
a compiler update that reorders them fails this test instead of racing on the LDS operands."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "unidepth_amd", "csrc", "gemm_pp.hip")


def _device_asm():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "gemm_pp.s")
        flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]      # = csrc/build.sh FLAGS
        subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", SRC, "-o", out], check=True, capture_output=True, timeout=600)
        return open(out).read()


_ASM = {}


@pytest.mark.parametrize("nw,n_dma", [(8, 11), (4, 14)])
def test_ping_pong_prologue_order_and_counted_wait(nw, n_dma):
    """nw = 8: the product's 8-wave form (A(0): 3 DMA instructions per thread); nw = 4: the two-workgroups-per-CU instantiation (A(0): 6)."""
    if "asm" not in _ASM:
        _ASM["asm"] = _device_asm()
    asm = _ASM["asm"]
    m = re.search(r"^(_ZN\S*gemm_pp_f32_kernelILi3ELi%dEE[^:\s]*):[^\n]*\n(.*?)\n\s*s_endpgm" % nw, asm, re.S | re.M)
    assert m, "gemm_pp_f32_kernel<3, %d> not found in the device assembly" % nw
    body = m.group(2).splitlines()
    ins = [l.strip() for l in body if l.strip() and not l.strip().startswith((";", ".", "//"))]
    first_wait = next(i for i, l in enumerate(ins) if l.startswith("s_waitcnt") and "vmcnt(28)" in l)
    vmem = [l.split()[0] + (" lds" if l.rstrip().endswith("lds") else "") for l in ins[:first_wait]
            if l.split()[0].startswith(("buffer_", "global_", "flat_", "scratch_"))]
    # the accumulate path (full tile): 11 (14) operand DMA instructions, then the 24 old-value loads, nothing else on the vector-memory queue
    tail = vmem[-(n_dma + 24):]
    assert tail == ["buffer_load_dwordx4 lds"] * n_dma + ["global_load_dwordx4"] * 24, tail
    assert not any(v.startswith(("global_store", "buffer_store", "scratch_")) for v in vmem), "stores / scratch traffic before the counted wait"
    # every other vector-memory load before that wait belongs to another path of the prologue (partial tile: guarded loads + vmcnt(0))
    assert all(v in ("buffer_load_dwordx4 lds", "global_load_dwordx4") for v in vmem), set(vmem)
    # the K loop's counted wait: W(kt + 2) (4 DMA instructions) stays in flight across the K-tile boundary
    assert sum(1 for l in ins if l.startswith("s_waitcnt") and re.search(r"vmcnt\(4\)", l)) >= 2
    # 8 waves x 256 VGPRs = the whole register file of a CU: no spills to scratch, at most 256 registers
    meta = re.search(r"\.name:\s+%s\n(.*?)\.wavefront_size" % re.escape(m.group(1)), asm, re.S)
    assert meta, "kernel metadata not found"
    assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta.group(1)).group(1)) == 0
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta.group(1)).group(1)) <= 256
    assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta.group(1)).group(1)) == 0
