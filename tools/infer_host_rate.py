import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import annonet_amd as aa
net = aa.RuntimeNet(aa.net_config(2, 3, 3, 1.0, 1, aa.ANH_BF16))
rng = np.random.default_rng(3)
from tests.conftest import random_params
from oracle.oracle import OracleNet
o = OracleNet(2, 3, 3, 1.0, 1)
p, r = random_params(o, 2)
net.set_params(p, r)
side = 4096
img = rng.integers(0, 256, (side, side, 3), dtype=np.uint8)
tp = aa.tiling.parameters(1024, 1024, 35, 35)
for _ in range(2): out = aa.annonet_infer(net, img, tiling_parameters=tp)
t0 = time.perf_counter()
for _ in range(5): out = aa.annonet_infer(net, img, tiling_parameters=tp)
dt = (time.perf_counter() - t0) / 5
print(f"annonet_infer host path 4096^2: {dt*1e3:.1f} ms, {side*side/dt/1e6:.0f} Mpx/s")
