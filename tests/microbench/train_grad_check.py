"""Gradients of one training-mode forward/backward against float64 (CPU): library path (NCHW), library path in channels-last,
and the native-conv path.  Prints max |g - g64| / max|g64| over all parameters and the worst parameter."""
import copy, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xiangqi_alphazero_amd import model, weights
if os.environ.get("NO_TF32"):
    torch.backends.cudnn.allow_tf32 = False; torch.backends.cuda.matmul.allow_tf32 = False
if os.environ.get("DETERMINISTIC"):
    torch.backends.cudnn.deterministic = True
C, B = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 3
net = model.XiangqiNet(C, B); net.load_state_dict(weights.make_state_dict(C, B, seed=9)); net.train()
gen = torch.Generator().manual_seed(4)
x = (torch.rand(96, 15, 10, 9, generator=gen) < 0.1).float(); pi = torch.softmax(torch.randn(96, 8100, generator=gen), 1); z = torch.rand(96, 1, generator=gen) * 2 - 1
def grads(m, dev, dt):
    m = copy.deepcopy(m).to(dev, dt)
    logits, value = m(x.to(dev, dt))
    loss = -torch.mean(torch.sum(pi.to(dev, dt) * F.log_softmax(logits, dim=1), dim=1)) + F.mse_loss(value, z.to(dev, dt))
    loss.backward()
    return loss.item(), {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()}
l64, g64 = grads(net, "cpu", torch.float64)
variants = {"library NCHW": net, "library channels-last": copy.deepcopy(net).to(memory_format=torch.channels_last)}
nat = copy.deepcopy(net).cuda(); nat.use_native_conv(True); variants["native conv"] = nat
for name, m in variants.items():
    l, g = grads(m, "cuda", torch.float32)
    worst = max(((g[n] - g64[n]).abs().max().item() / g64[n].abs().max().item(), n) for n in g64)
    print("%-24s loss err %.2e   worst grad err / max|grad| %.2e (%s)" % (name, abs(l - l64) / abs(l64), worst[0], worst[1]))
