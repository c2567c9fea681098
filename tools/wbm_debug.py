#!/usr/bin/env python3
"""Debug aid for conv_wino_bf16m.hip: one 3x3 layer (act none) against torch conv2d in float64; prints where the error sits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsic_amd import ops
B = int(os.environ.get("B", "1")); H = int(os.environ.get("H", "64")); W = int(os.environ.get("W", "64"))
Cin = int(os.environ.get("CIN", "128")); Cout = int(os.environ.get("COUT", "128"))
mode = os.environ.get("LAYER", "3x3")
g = torch.Generator().manual_seed(1)
x = (torch.rand((B, Cin, H, W), generator=g) * 2 - 1)
if os.environ.get("ONEHOT"):   # a single non-zero input pixel / channel
    x.zero_(); x[0, int(os.environ.get("C0", "0")), int(os.environ.get("Y0", "5")), int(os.environ.get("X0", "7"))] = 1.0
bias = torch.zeros(Cout)
ACT = os.environ.get("ACT", "none")
beta = (0.5 + torch.rand(Cout, generator=g)); gamma = (0.02 + 0.28 * torch.rand(Cout, generator=g))
if ACT != "none": bias = torch.rand(Cout, generator=g) - 0.5
code = {"none": ops.ACT_NONE, "gdn": ops.ACT_GDN, "igdn": ops.ACT_IGDN, "relu": ops.ACT_RELU}[ACT]
def act_ref(r):
    r = r + bias.double().view(1, -1, 1, 1)
    if ACT == "relu": return torch.relu(r)
    if ACT in ("gdn", "igdn"):
        d = torch.sqrt(beta.double().view(1, -1, 1, 1) + gamma.double().view(1, -1, 1, 1) * r * r)
        return r * d if ACT == "igdn" else r / d
    return r
if mode == "3x3":
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * 0.05
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    u = ops.split_wino_weight_bf16(ops.pack_wino_weight(w.cuda()), Cout, Cin)
    y = ops.conv3x3_wino_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), u, bias.cuda(), Cout, code, beta.cuda(), gamma.cuda())
elif mode == "s2":
    Cs = Cin
    x = (torch.rand((B, Cs, 2 * H, 2 * W), generator=g) * 2 - 1)
    w = (torch.rand((Cout, Cs, 5, 5), generator=g) * 2 - 1) * 0.05
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=2)
    xs = ops.space_to_depth(x.permute(0, 2, 3, 1).contiguous().cuda())
    u = ops.split_wino_weight_bf16(ops.pack_wino_s2_weight(w.cuda()), Cout, 4 * Cs)
    y = ops.conv3x3_wino_nhwc(xs, u, bias.cuda(), Cout, code, beta.cuda(), gamma.cuda(), s2d_in=True)
else:
    w = (torch.rand((Cin, Cout, 5, 5), generator=g) * 2 - 1) * 0.05
    ref = torch.nn.functional.conv_transpose2d(x.double(), w.double(), stride=2, padding=2, output_padding=1)
    u = ops.split_wino_weight_bf16(ops.pack_wino_convT_weight(w.cuda()), Cout, Cin, 4)
    y = ops.conv_transpose2d_wino_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), u, bias.cuda(), Cout, code, beta.cuda(), gamma.cuda())
got = y.permute(0, 3, 1, 2).double().cpu()
ref = act_ref(ref)
err = (got - ref).abs()
print(f"{mode} B={B} {H}x{W} Cin={Cin} Cout={Cout}: max err {err.max():.3e}  |ref|max {ref.abs().max():.3e}")
if err.max() > 1e-3 * ref.abs().max():
    OH, OW = err.shape[-2:]
    e = err.amax(dim=(0, 1))                        # [OH, OW]
    T = 32 if mode == "convT" else 16
    em = e.reshape(OH // T, T, OW // T, T).amax(dim=(0, 2))
    torch.set_printoptions(precision=1, linewidth=250, sci_mode=False)
    print("max err by pixel position inside a workgroup tile (rows = y):")
    print((em / ref.abs().max()).clamp(max=9.9))
    ec = err.amax(dim=(0, 2, 3)).reshape(-1, 32).amax(dim=1)
    print("max err by 32-channel group:", (ec / ref.abs().max()).tolist())
    et = e.reshape(OH // T, T, OW // T, T).amax(dim=(1, 3))
    print("max err by workgroup tile:"); print((et / ref.abs().max()).clamp(max=9.9))
if os.environ.get("ONEHOT"):
    d = (got - ref)[0]
    nz = (d.abs().amax(dim=0) > 1e-6).nonzero()
    print("wrong pixels (y, x):", nz.tolist()[:40])
    gz = (got[0].abs().amax(dim=0) > 1e-6).nonzero()
    rz = (ref[0].abs().amax(dim=0) > 1e-6).nonzero()
    print("got nonzero at:", gz.tolist()[:40]); print("ref nonzero at:", rz.tolist()[:40])
    for (yy, xx) in gz.tolist()[:12]:
        print((yy, xx), "got", got[0, :3, yy, xx].tolist(), "ref", ref[0, :3, yy, xx].tolist())
if os.environ.get("PIX"):
    yy, xx = [int(v) for v in os.environ["PIX"].split(",")]
    raw = ref  # activated reference
    print("pixel", (yy, xx), "got", got[0, :4, yy, xx].tolist())
    print("  ref", ref[0, :4, yy, xx].tolist())
    for (dy, dx) in [(0, -2), (0, -1), (0, 1), (-1, 0), (1, 0), (-2, 0)]:
        print("  ref at", (yy + dy, xx + dx), ref[0, :4, yy + dy, xx + dx].tolist())
if B > 1:
    eb = err.amax(dim=(1, 2, 3)); print("max err per image:", eb.tolist())
    b = int(eb.argmax()); e2 = err[b].amax(dim=0); pos = (e2 > 1e-2).nonzero()
    print("image", b, "wrong pixels:", len(pos), pos[:12].tolist())
    if len(pos):
        yy, xx = pos[0].tolist()
        print(" got", got[b, :6, yy, xx].tolist()); print(" ref", ref[b, :6, yy, xx].tolist())
        ch = (err[b, :, yy, xx] > 1e-2).nonzero().flatten().tolist(); print(" wrong channels at that pixel:", ch[:40], len(ch))
        for c in ch[:4]:
            print(f"  ch {c}: got {got[b, c, yy, xx]:.6f} ref {ref[b, c, yy, xx]:.6f}; same channel, same image: ref at x-2 {ref[b, c, yy, xx-2]:.6f}, y-16 {ref[b, c, yy-16, xx] if yy >= 16 else 0:.6f}; other image ref {ref[1-b, c, yy, xx]:.6f}")
        # search where got's value appears in ref
        val = got[b, ch[0], yy, xx]
        hit = ((ref - val).abs() < 2e-4 * max(1.0, abs(float(val)))).nonzero()
        print("  got value found in ref at:", hit[:8].tolist())
if os.environ.get("HIST"):
    import collections
    bad = (err > 1e-2 * ref.abs().max()).nonzero()   # [n, c, y, x]
    T = 32 if mode == "convT" else 16
    hist = collections.Counter()
    for n_, c_, y_, x_ in bad.tolist():
        ty, i = (y_ % 16) // 2, y_ % 2
        tx, j = (x_ % 16) // 2, x_ % 2
        t = ty * 8 + tx; m = t // 32; r = t % 32
        h = (r >> 2) & 1; e = (r & 3) + 4 * (r >> 3)
        hist[(f"nt{c_ // 32}", f"pq{i}", f"j{j}", f"m{m}", f"e{e}", f"h{h}", f"l{c_ % 32}")] += 1
    print("wrong elements:", len(bad)); 
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:40]: print("  ", k, v)
