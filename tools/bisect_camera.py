import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, warnings
warnings.simplefilter("ignore")
from oracle import cases, restate, synth
from unidepth_amd import UniDepthV2
for name in ("vits_462x616_b1", "vitb_518x518_b1", "vitl_518x518_b1"):
    case = cases.CASES[name]
    cfg = synth.load_config(case["arch"]); sd = synth.make_synthetic_checkpoint(cfg, case["ckpt_seed"])
    rgb, cam = cases.case_inputs(case)
    ref = restate.OracleV2(cfg, sd).infer(rgb, None)
    model = UniDepthV2(cfg).load_state_dict(sd).to("cuda").eval()
    out = model.infer(rgb.cuda(), None)
    out2 = model.infer(rgb.cuda(), ref["intrinsics"][0])       # oracle's own predicted K fed back as GT camera
    torch.cuda.synchronize()
    f = lambda o: ((o["depth"].cpu() - ref["depth"]).abs() / ref["depth"]).mean().item()
    kerr = ((out["intrinsics"].cpu() - ref["intrinsics"]).abs() / ref["intrinsics"].abs().clamp_min(1)).max().item()
    print(f"{name}: predicted-camera depth ARel {f(out):.3e} (K max-rel {kerr:.2e}) | oracle-K rays depth ARel {f(out2):.3e}")
