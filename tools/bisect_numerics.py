import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, warnings
warnings.simplefilter("ignore")
from oracle import cases, restate, synth
from unidepth_amd import UniDepthV2, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "vits_462x616_b1"
case = cases.CASES[name]
cfg = synth.load_config(case["arch"]); sd = synth.make_synthetic_checkpoint(cfg, case["ckpt_seed"])
rgb, cam = cases.case_inputs(case)
ref = restate.OracleV2(cfg, sd).infer(rgb, cam)
_lib.lib.ud_set_debug_flags.argtypes = [ctypes.c_int]
for flags in (0, 1, 2, 4, 8, 15):
    _lib.lib.ud_set_debug_flags(flags)
    model = UniDepthV2(cfg).load_state_dict(sd).to("cuda").eval()
    out = model.infer(rgb.cuda(), cam)
    torch.cuda.synchronize()
    d = ((out["depth"].cpu() - ref["depth"]).abs() / ref["depth"]).mean().item()
    f = ((out["depth_features"].cpu() - ref["depth_features"]).norm() / ref["depth_features"].norm()).item()
    print(f"flags={flags:2d}  depth_arel={d:.3e}  feat_rel={f:.3e}")
