#!/usr/bin/env python3
"""gs_match_template 128x128 on 1280x720 and 3840x2160 for rocprofv3 passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from oracle.pyoracle import Oracle
g = gs.lib(); g.use_torch_stream()
for (iw, ih) in ((1280, 720), (3840, 2160)):
    img = Oracle.synth(iw, ih, 4); d_img = torch.from_numpy(img).cuda()
    tw = th = 128
    t = torch.from_numpy(img[100:100 + th, 200:200 + tw].copy()).cuda()
    r = torch.zeros((ih - th + 1, iw - tw + 1), dtype=torch.uint8, device="cuda")
    for _ in range(3): g.match_template(d_img, t, r)
torch.cuda.synchronize()
