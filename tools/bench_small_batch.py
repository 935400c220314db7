"""infer() latency at small batches (ViT-L/14, 518x518, batch 1 / 2 / 4), one call at a time, with the 128x128 GEMM kernel's
variants for tile counts below the CU count (pipelined 4-stage ring + two-way K split = default, ring without the split = debug bit 32, plain 2-stage kernel = bit 16) in one process -- the A/B for the small-batch path."""
import ctypes
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle import synth  # noqa: E402  (synthetic checkpoint only; nothing of the oracle is timed)
from unidepth_amd import UniDepthV2, _lib  # noqa: E402


def main():
    cfg = synth.load_config("vitl14")
    model = UniDepthV2(cfg).load_state_dict(synth.make_synthetic_checkpoint(cfg, 1)).to("cuda").eval()
    res = {}
    for B in (1, 2, 4):
        rgb = torch.randint(0, 256, (B, 3, 518, 518), dtype=torch.uint8, device="cuda")
        for rnd in range(2):
            for flag in (0, 32, 16):
                _lib.lib.ud_set_debug_flags(ctypes.c_int(flag))
                for _ in range(5):
                    model.infer(rgb)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 40
                for _ in range(n):
                    model.infer(rgb)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / n * 1e3
                res.setdefault(f"b{B}_{ {0: 'ring4_ksplit', 32: 'ring4', 16: 'plain'}[flag] }", []).append(round(ms, 3))
    _lib.lib.ud_set_debug_flags(ctypes.c_int(0))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
