"""The sampling harness end to end on the GPU in --synthetic mode (random full-size FLUX weights, synthetic MLLM
hidden states): every launch goes through the HIP path; outputs are packed latents saved per job."""
import glob
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_qwenvl_harness_synthetic(tmp_path):
    from x2i_amd.infer import inference_qwenvl
    inference_qwenvl.main(["--synthetic", "--qwen_size", "3b", "--task", "image2image", "--height", "256", "--width", "256",
                           "--batch", "2", "--outputs", str(tmp_path), "--seed", "5"])
    files = sorted(glob.glob(os.path.join(str(tmp_path), "image2image", "*_latents.pt")))
    assert len(files) == 2
    lat = torch.load(files[0])
    assert lat.shape == (2, 256, 64) and lat.dtype == torch.bfloat16 and torch.isfinite(lat.float()).all()
    assert lat.float().std() > 0.1


def test_qwenvl_harness_synthetic_with_vae_decode(tmp_path):
    from x2i_amd.infer import inference_qwenvl
    inference_qwenvl.main(["--synthetic", "--decode", "--qwen_size", "7b", "--task", "video2image", "--height", "256", "--width", "256",
                           "--outputs", str(tmp_path), "--num_steps", "2"])
    files = sorted(glob.glob(os.path.join(str(tmp_path), "video2image", "*.jpg")))
    assert len(files) == 2
    from PIL import Image
    assert Image.open(files[0]).size == (256, 256)
