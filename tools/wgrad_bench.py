"""Weight-gradient micro-benchmark (mixed precision): the dominant SPADE gamma|beta shape of train_generator.
env: CIN, COUT, K, H, W, N, XBF (x stored bf16), YBF (dy stored bf16), ROUNDS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hr_viton_amd  # noqa
from hr_viton_amd import ops, train_ops as T

E = lambda k, d: int(os.environ.get(k, d))
N, H, W, cin, cout, k = E("N", 4), E("H", 1024), E("W", 768), E("CIN", 128), E("COUT", 128), E("K", 3)
xbf, ybf, rounds = E("XBF", 1), E("YBF", 1), E("ROUNDS", 5)
T.MMA_BF16[0] = E("MIXED", 1) == 1
dt = lambda b: torch.bfloat16 if b else torch.float32
x = ops.Act(torch.randn(N, H, W, cin, device="cuda").to(dt(xbf)), cin)
dy = ops.Act(torch.randn(N, H, W, cout, device="cuda").to(dt(ybf)), cout)
dw = torch.empty(cout, cin, k, k, device="cuda")
db = torch.empty(cout, device="cuda")
for _ in range(2):
    T.conv_wgrad(dy, x, 0, 0, cin, k, k, 1, k // 2, dw, dbias=db)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(rounds):
    T.conv_wgrad(dy, x, 0, 0, cin, k, k, 1, k // 2, dw, dbias=db)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / rounds
print(f"wgrad N={N} {H}x{W} cin={cin} cout={cout} k={k} xbf={xbf} ybf={ybf}: {ms:.3f} ms  {2.0*N*H*W*cin*cout*k*k/ms/1e9:.1f} TFLOP/s")
