import torch, sys
sys.path.insert(0, ".")
from unidepth_amd import ops
def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()
B,Hin,Win,Cin,Co,k=2,5,7,64,64,1
rows_in=40
x=rnd(B,rows_in,Cin,seed=1).half()
wt=rnd(Cin,Co,k,k,scale=Cin**-0.5,seed=2)
Wg=wt.permute(2,3,1,0).reshape(k*k*Co,Cin).half()
bias=torch.zeros(Co,device="cuda")
lat=torch.zeros(B,Hin*Win,Co,device="cuda")
ops.gemm(A=x, W=Wg, bias=bias, out=lat, M=B*rows_in, N=k*k*Co, K=Cin, lda=Cin, ldw=Cin, ldc=Co, ldc2=Co, epi=ops.UD_EPI_D2S,
         d2s_k=k, d2s_Co=Co, d2s_Hin=Hin, d2s_Win=Win, d2s_rows_in_img=rows_in, d2s_out_img_pix=Hin*Win)
torch.cuda.synchronize()
full=(x.float().view(-1,Cin)@Wg.float().t())
got=lat.view(-1,Co)
print("got rows", got.shape, "nonzero rows", (got.abs().sum(1)>0).sum().item())
# match rows
for r in [0,1,2,7,34,35,36,69]:
    d=((full-got[r])**2).sum(1); j=d.argmin().item()
    print("got row",r,"best full row",j,"err",d[j].item(), "| same-row err", ((full[r+ (5 if r>=35 else 0)]-got[r])**2).sum().item())
# column check for row 0
r0=full[0]; g0=got[0]
print("row0 full", r0[:8]); print("row0 got ", g0[:8])
for c in range(8):
    j=(r0-g0[c]).abs().argmin().item(); print("got col",c,"matches full col",j, (r0[j]-g0[c]).abs().item())
