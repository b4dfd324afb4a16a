import subprocess, sys, time, threading
import torch
sys.path.insert(0, ".")
from xiangqi_alphazero_amd import hip
B, C = 8192, 256
x = torch.randn(B, 90, C, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda") * 0.02
u = hip.wino_transform_weights(w); bias = torch.randn(C, device="cuda"); res = torch.randn(B, 90, C, device="cuda")
y = torch.empty_like(x)
L = hip.lib()
def run():
    L.xq_wino_conv3x3(x.data_ptr(), u.data_ptr(), bias.data_ptr(), res.data_ptr(), y.data_ptr(), B, C, 1, hip.stream_ptr(x.device))
def smi(tag):
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "junction" in l.lower() or "mclk" in l]
    print(tag, " | ".join(keep), flush=True)
smi("idle")
for rep in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): run()
    e1.record()
    time.sleep(0.5)
    smi("busy%d" % rep)
    torch.cuda.synchronize()
    print("rep %d: %.3f ms/launch" % (rep, e0.elapsed_time(e1) / 300), flush=True)
