#!/usr/bin/env python3
"""gs_match_template, 128 x 128 template on a 3840 x 2160 frame, 20 calls: for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, grayskull_amd as gs
from oracle.pyoracle import Oracle
g = gs.lib(); g.use_torch_stream()
img = Oracle.synth(3840, 2160, 4)
t = img[100:228, 200:328].copy(); t[::3, ::5] ^= 0x55
d_img, d_t = torch.from_numpy(img).cuda(), torch.from_numpy(t).cuda()
r = torch.zeros((2160 - 127, 3840 - 127), dtype=torch.uint8, device="cuda")
for _ in range(20):
    g.match_template(d_img, d_t, r)
torch.cuda.synchronize()
