import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.lib(); g.use_torch_stream()
casc = Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin"))
n, h, w = 4, 1080, 1920
src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 3)
ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(src, ii)
dc = g.cascade_create(casc)
rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
g.tune(4, int(os.environ.get("LBP_PRESET", 0)))
for _ in range(2): g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1)
torch.cuda.synchronize()
