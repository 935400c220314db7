#!/usr/bin/env python
"""A/B of attention-kernel variant libraries (tools/attn_variants.sh -> ab/libattn_<name>.so) on the encoder shape
(B=8, H=16, N=1370, q pre-scaled, MODE 1).  One subprocess per (library, round), rounds interleaved so that box drift hits every
variant alike; the child also checks the result against an fp32 torch softmax(QK^T)V on two (image, head) pairs, one of them with
a SPIKED key row placed in a late tile (forces the rare rescale path: cdna guide rule 26).  GPU box only.
usage: python tools/attn_ab.py [--rounds 3] name1 name2 ...      (names of ab/libattn_<name>.so; 'product' = the in-tree library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from unidepth_amd import ops
    B, H, N = 8, 16, 1370
    D = H * 64; Np = 1376; kvld = 1408
    g = torch.Generator().manual_seed(0)
    c = 0.125 * 1.4426950408889634
    q = torch.randn(B * Np, D, generator=g) * 2.0
    k = torch.randn(B * Np, D, generator=g) * 2.0
    # spike: key 1000 of (image 0, head 1) aligned with query 5 -> that row's maximum jumps by far more than 2^15 at tile 15
    k[1000, 64:128] = q[5, 64:128] * 6.0
    k[3 * Np + 700, 7 * 64:8 * 64] = q[3 * Np + 9, 7 * 64:8 * 64] * 0.9      # a moderate spike on (img 3, head 7): P ~ 2^23 against the stale offset, finite in fp32
    qk = torch.cat([q * c, k], dim=1).half().cuda()                     # Q stored pre-scaled by scale * log2(e), as the engine does
    v = torch.randn(B, H, N, 64, generator=g)
    cols = ((torch.arange(N) & ~15) | ((torch.arange(N) & 4) << 1) | ((torch.arange(N) & 8) >> 1) | (torch.arange(N) & 3))
    vt = torch.zeros(B, H, 64, kvld)
    vt[:, :, :, cols] = v.permute(0, 1, 3, 2)
    vt = vt.half().cuda()
    o = torch.zeros(B * Np, D, dtype=torch.half, device="cuda")
    P = ops.Program()
    P.attention(Q=qk, K=qk.data_ptr() + D * 2, Vt=vt, O=o, B=B, H=H, Nq=N, Nk=N, ldq=2 * D, ldk=2 * D, ldo=D, kv_ld=kvld,
                q_rows_per_img=Np, k_rows_per_img=Np, scale=0.125, q_prescaled=1)
    for _ in range(5):
        P.run()
    torch.cuda.synchronize()
    # ---- correctness on (img 0, head 1) [spiked] and (img 3, head 7)
    worst = 0.0
    for img, hd in ((0, 1), (3, 7)):
        qh = qk[img * Np: img * Np + N, hd * 64:(hd + 1) * 64].float() / c * 0.125
        kh = qk[img * Np: img * Np + N, D + hd * 64: D + (hd + 1) * 64].float()
        vh = vt[img, hd][:, cols].t().float()
        ref = torch.softmax(qh @ kh.t(), dim=-1) @ vh
        got = o[img * Np: img * Np + N, hd * 64:(hd + 1) * 64].float()
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, err)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(30):
            P.run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 30 * 1e3)
    print(json.dumps({"us": min(ts), "us_all": ts, "max_rel_err": worst}))


def main():
    rounds = 3
    names = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--rounds" in sys.argv:
        rounds = int(sys.argv[sys.argv.index("--rounds") + 1]); names.remove(str(rounds))
    res = {n: [] for n in names}
    errs = {}
    for r in range(rounds):
        for n in (names if r % 2 == 0 else names[::-1]):
            env = dict(os.environ)
            if n != "product":
                env["UNIDEPTH_HIP_LIB"] = os.path.join(ROOT, "ab", f"libattn_{n}.so")
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                res[n].append(d["us"]); errs[n] = d["max_rel_err"]
            except Exception:
                print(f"{n}: FAILED\n{out.stdout[-400:]}\n{out.stderr[-800:]}")
    fl = 4.0 * 8 * 16 * 1370 * 1370 * 64
    for n in names:
        if res[n]:
            best = min(res[n])
            print(f"{n:18s} best {best:6.1f} us ({fl / best / 1e6:6.0f} TFLOP/s)  rounds " + " ".join(f"{x:6.1f}" for x in res[n]) + f"   max rel err {errs[n]:.2e}")


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
