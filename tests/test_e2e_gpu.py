"""End-to-end tolerance of SURVEY.md section 8(d): projector-free 4-step sampling + latent post-processing + VAE decode on the
HIP path vs the same chain in the CPU oracle (fp32 math on the same bf16-rounded weights): final latents rel-L2 <= 5e-2 and
decoded-image PSNR >= 30 dB."""
import math

import pytest
import torch

from oracle import flux as OF
from oracle import sampler as OS
from oracle import vae as OV
from tests.util import golden, rel_l2, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_four_step_sampling_then_decode_psnr():
    from x2i_amd.flux import FluxTransformer2DModel
    from x2i_amd.pipeline import FluxPipeline, FlowMatchEulerDiscreteScheduler
    from x2i_amd.vae import AutoencoderKL
    _, meta = golden("flux_tiny_schnell")
    cfg = meta["cfg"]
    sd = OF.random_flux_state_dict(cfg, seed=meta["weight_seed"], std=meta["weight_std"])
    m = FluxTransformer2DModel(**cfg, device=DEV)
    m.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    vcfg = dict(OV.FLUX_VAE_CFG, block_out_channels=(128, 128, 256, 256))
    vsd = OV.random_vae_decoder_state_dict(vcfg, seed=11)
    vae = AutoencoderKL(block_out_channels=vcfg["block_out_channels"], device=DEV)
    vae.load_state_dict({k: v.to(torch.bfloat16) for k, v in vsd.items()}, strict=True)

    H, W, B = 128, 192, 2
    pipe = FluxPipeline(m, FlowMatchEulerDiscreteScheduler(**OS.SCHEDULER_SCHNELL))
    pe, pooled = seeded((B, 40, 128), 5).bfloat16(), seeded((B, 64), 6).bfloat16()
    noise = OS.pack_latents(torch.randn((B, 16, H // 8, W // 8), generator=torch.Generator().manual_seed(0))).bfloat16()
    lat_hip = pipe(prompt_embeds=pe.to(DEV), pooled_prompt_embeds=pooled.to(DEV), num_inference_steps=4, guidance_scale=3.5,
                   height=H, width=W, output_type="latent", latents=noise.to(DEV)).images
    # reference harness post-processing (infer/inference_qwenvl.py:209-214): unpack, /scaling_factor + shift_factor, decode
    vsf = 2 ** len(vae.config.block_out_channels)
    z_hip = FluxPipeline._unpack_latents(lat_hip, H, W, vsf) / vae.config.scaling_factor + vae.config.shift_factor
    img_hip = vae.decode(z_hip, return_dict=False)[0].float().cpu()

    sdr = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    lat = noise.clone()
    ts, sig = OS.flow_match_sigmas(4, OS.SCHEDULER_SCHNELL, lat.shape[1])
    img_ids, txt_ids = OS.prepare_latent_image_ids(H // 16, W // 16), torch.zeros(40, 3)
    for i, tt in enumerate(ts):
        t1000 = ((tt.expand(B).to(torch.bfloat16) / 1000) * 1000).float()
        eps = OF.flux_forward(sdr, cfg, lat.float(), pe.float(), pooled.float(), t1000 / 1000, img_ids, txt_ids)
        lat = OS.euler_step(lat, eps.bfloat16(), sig[i], sig[i + 1])
    assert rel_l2(lat_hip, lat) < 5e-2
    z_ref = (OS.unpack_latents(lat, H, W, vsf) / vcfg["scaling_factor"] + vcfg["shift_factor"]).to(torch.bfloat16).float()
    img_ref = OV.vae_decode({k: v.to(torch.bfloat16).float() for k, v in vsd.items()}, z_ref, vcfg)
    assert img_hip.shape == img_ref.shape == (B, 3, H, W)
    peak = float(img_ref.max() - img_ref.min())
    mse = float(((img_hip - img_ref) ** 2).mean())
    psnr = 10 * math.log10(peak * peak / mse)
    assert psnr >= 30.0, psnr
