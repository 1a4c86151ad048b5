#!/usr/bin/env python3
"""Time the FLUX VAE decode (N1) at 1024x1024 on the HIP path."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd.vae import AutoencoderKL
B = int(os.environ.get("X2I_B", "4"))
vae = AutoencoderKL(device="cuda").init_random_(0)
z = torch.randn(B, 16, 128, 128, device="cuda").bfloat16()
for _ in range(2):
    img = vae.decode(z, return_dict=False)[0]
torch.cuda.synchronize()
t = time.time()
for _ in range(3):
    img = vae.decode(z, return_dict=False)[0]
torch.cuda.synchronize()
dt = (time.time() - t) / 3
print(f"vae decode B={B} 1024^2: {dt*1e3:.1f} ms  ({dt/B*1e3:.1f} ms/image)  out {tuple(img.shape)} finite={bool(torch.isfinite(img.float()).all())}")
