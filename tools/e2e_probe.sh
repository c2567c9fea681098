cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
from dsic_amd import synthetic as S, metrics, entropy
from dsic_amd.model import CompressionModel
B,H,W,C=64,256,256,3
m = CompressionModel(N=128, M=192, spatial_params=False, min_nu=2, max_nu=100.0, in_ch=C)
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.make_state_dict(seed=1).items()}, strict=True)
m = m.cuda().eval()
x = torch.from_numpy(S.make_patches(0, B, H, W, C)).cuda()
u8 = (x.permute(0, 2, 3, 1).clamp(0, 1) * 255.0).round().to(torch.uint8).contiguous()
host_in = u8.cpu().pin_memory()
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
print("forward f32 resident      ", t(lambda: m(x, quant_mode="round")))
print("forward u8 resident       ", t(lambda: m(u8, quant_mode="round")))
d = torch.empty_like(u8)
print("H2D 12.6 MB pinned        ", t(lambda: d.copy_(host_in, non_blocking=True)))
hb = torch.empty((B, 102440), dtype=torch.uint8).pin_memory(); db = torch.zeros((B,102440), dtype=torch.uint8, device="cuda")
print("D2H 6.5 MB pinned         ", t(lambda: hb.copy_(db, non_blocking=True)))
print("u8->f32 nchw (torch)      ", t(lambda: u8.permute(0, 3, 1, 2).float() / 255.0))
xf = u8.permute(0, 3, 1, 2).float() / 255.0
o = m(x, quant_mode="round")
print("ms-ssim on non-contig ref ", t(lambda: metrics.ms_ssim_per_image(o["x_hat"], xf, clamp_x=True)))
print("ms-ssim on contiguous ref ", t(lambda: metrics.ms_ssim_per_image(o["x_hat"], x, clamp_x=True)))
PY
