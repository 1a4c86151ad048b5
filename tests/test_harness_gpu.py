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


def test_real_prompt_batches_match_one_prompt_at_a_time(tmp_path):
    """Row H batching: five prompts of two different text lengths through generate_jobs (packed per length into batched sampling
    calls) give exactly the latents of five B = 1 calls with the same global noise -- i.e. what the reference's one-prompt loop
    computes -- because a sample's result is bit-independent of its batch."""
    from x2i_amd.infer import harness as H
    from x2i_amd.infer.inference_qwenvl import tasks  # noqa: F401  (import check)
    args = H.build_parser("minicpm").parse_args(["--synthetic", "--height", "256", "--width", "256", "--outputs", str(tmp_path),
                                                 "--batch", "4", "--seed", "3", "--num_steps", "2"])
    lens = {"a": 96, "b": 40, "c": 96, "d": 40, "e": 96}

    class Cond:  # stands in for the MLLM: deterministic hidden states per prompt, unpadded (MiniCPM-style) lengths
        def __call__(self, text_prompt=None, **kw):
            g = torch.Generator(device="cuda").manual_seed(sum(map(ord, text_prompt)))
            return (torch.randn((1, 29, lens[text_prompt], 3584), device="cuda", generator=g) * 3).bfloat16()

    h = H.Harness(args, "minicpm", Cond(), "cuda")
    args.synthetic = False  # weights stay synthetic; the run path is the real-prompt one
    jobs = [dict(filename=k, text_prompt=k) for k in lens]
    h.run_tasks({"text2image": jobs})
    files = sorted(glob.glob(os.path.join(str(tmp_path), "text2image", "*_latents.pt")))
    assert [os.path.basename(f) for f in files] == ["%s_0_latents.pt" % k for k in lens]
    # first chunk = 4 prompts (lengths 96, 40, 96, 40 -> two sampling calls of B = 2), second chunk = 1 prompt
    C = h.pipeline.transformer.config.in_channels // 4
    for chunk in (list(lens)[:4], list(lens)[4:]):
        noise, _ = h.pipeline.prepare_latents(len(chunk), C, 256, 256, torch.bfloat16, h.device, torch.Generator("cuda").manual_seed(3))
        for i, k in enumerate(chunk):
            pooled, embeds = h.embeds(text_prompt=k)
            one = h.pipeline(prompt_embeds=embeds, pooled_prompt_embeds=pooled, num_inference_steps=2, guidance_scale=3.5, height=256,
                             width=256, output_type="latent", latents=noise[i:i + 1]).images
            got = torch.load(os.path.join(str(tmp_path), "text2image", "%s_0_latents.pt" % k))
            assert torch.equal(got, one.cpu()), k


def test_multi_turn_synthetic_lengths_and_graph_policy(tmp_path, capsys):
    """infer/inference_multi_turn.py:132-156: the text length grows with the conversation.  --synthetic walks a list of lengths; a new length
    costs ONE eager pass, a returning one is captured once and replayed afterwards (the pipeline's graph statistics are printed per turn),
    and a repeated turn (same length, same seed-0 noise, same synthetic conditioning call index aside) writes finite latents."""
    from x2i_amd.infer import inference_multi_turn
    inference_multi_turn.main(["--synthetic", "--qwen_size", "3b", "--height", "256", "--width", "256", "--num_steps", "2",
                               "--turn_lengths", "48,80,48,48,80", "--outputs", str(tmp_path)])
    out = [l for l in capsys.readouterr().out.splitlines() if l.startswith("turn ")]
    assert len(out) == 5
    assert "2 eager, 0 captures, 0 replays" in out[1]      # two new lengths: two eager passes
    assert "2 eager, 1 captures, 1 replays" in out[2]      # 48 comes back: captured, replayed
    assert "2 eager, 1 captures, 2 replays" in out[3]      # ... replay only
    assert "2 eager, 2 captures, 3 replays" in out[4]      # 80 comes back
    files = sorted(glob.glob(os.path.join(str(tmp_path), "multi_turn", "turn_*_latents.pt")))
    assert len(files) == 5
    for f in files:
        lat = torch.load(f)
        assert lat.shape == (1, 256, 64) and torch.isfinite(lat.float()).all()
