"""Per-parameter gradient error of the native train step against float64, and the weight-gradient kernel alone on the tensors the
step really feeds it (captured with hooks) against float64 of the same float32 tensors."""
import copy, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import model, weights, hip, native_conv
C, B = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 3
net = model.XiangqiNet(C, B); net.load_state_dict(weights.make_state_dict(C, B, seed=9)); net.train()
gen = torch.Generator().manual_seed(4)
x = (torch.rand(96, 15, 10, 9, generator=gen) < 0.1).float(); pi = torch.softmax(torch.randn(96, 8100, generator=gen), 1); z = torch.rand(96, 1, generator=gen) * 2 - 1
def grads(m, dev, dt):
    logits, value = m(x.to(dev, dt))
    loss = -torch.mean(torch.sum(pi.to(dev, dt) * F.log_softmax(logits, dim=1), dim=1)) + F.mse_loss(value, z.to(dev, dt))
    loss.backward()
    return {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()}
g64 = grads(copy.deepcopy(net).double(), "cpu", torch.float64)
nat = copy.deepcopy(net).cuda().use_native_conv(True)
cap = {}
orig = native_conv.hip.wino_wgrad
def spy(xv, gv):
    cap[len(cap)] = (xv.clone(), gv.clone())
    return orig(xv, gv)
native_conv.hip.wino_wgrad = spy
g = grads(nat, "cuda", torch.float32)
for n in g64:
    e = (g[n] - g64[n]).abs().max().item() / g64[n].abs().max().item()
    if e > 1e-5:
        print("%-34s err/max %.2e   max|grad| %.3e" % (n, e, g64[n].abs().max().item()))
print("captured wgrad calls:", len(cap))
for k, (xv, gv) in cap.items():
    dw = orig(xv, gv).double()
    b = xv.shape[0]
    x64 = xv.double().view(b, 10, 9, C).permute(0, 3, 1, 2); g64_ = gv.double().view(b, 10, 9, C).permute(0, 3, 1, 2)
    w = torch.zeros(C, C, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    F.conv2d(x64, w, None, padding=1).backward(g64_)
    terms = F.conv2d(x64.abs(), torch.ones_like(w), None, padding=1)  # unused, scale only
    print("call %d: kernel err/max %.2e  max|dw| %.3e  |x|max %.2e |dy|max %.2e  sum|dy||x| scale %.3e" % (
        k, (dw - w.grad).abs().max().item() / w.grad.abs().max().item(), w.grad.abs().max().item(), xv.abs().max().item(), gv.abs().max().item(),
        (gv.abs().sum(dim=(0, 1)).max() * xv.abs().max()).item()))

# --- the convolution kernel (forward / data gradient) alone on the tensors of the real step
cap2 = []
orig_conv = native_conv._conv
def spy2(xv, u):
    out = orig_conv(xv, u)
    cap2.append((xv.clone(), u, out.clone()))
    return out
native_conv._conv = spy2
native_conv.hip.wino_wgrad = orig
nat2 = copy.deepcopy(net).cuda().use_native_conv(True)
grads(nat2, "cuda", torch.float32)
convs = [m for blk in nat2.res_blocks for m in (blk.conv1, blk.conv2)]
nf = len(convs)
for k, (xv, u, out) in enumerate(cap2):
    b = xv.shape[0]
    if k < nf:
        w = convs[k].weight.detach().double()
    else:                                                     # backward visits the convolutions in reverse order
        w = convs[nf - 1 - (k - nf)].weight.detach().double().flip(2, 3).transpose(0, 1)
    ref = F.conv2d(xv.double().view(b, 10, 9, C).permute(0, 3, 1, 2), w, None, padding=1)
    got = out.double()
    e = (got - ref).abs()
    print("conv call %2d (%s): max err / max|y| %.2e   max err of per-channel sums / max|sum| %.2e   max|y| %.2e  rms|y| %.2e" % (
        k, "fwd" if k < nf else "dgrad", e.max().item() / ref.abs().max().item(),
        (got.sum(dim=(0, 2, 3)) - ref.sum(dim=(0, 2, 3))).abs().max().item() / ref.sum(dim=(0, 2, 3)).abs().max().item(),
        ref.abs().max().item(), ref.pow(2).mean().sqrt().item()))
