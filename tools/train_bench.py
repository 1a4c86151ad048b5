"""Attention-distillation step (row N4) at the reference's training shape: FLUX.1-dev-sized transformer (19 + 38 blocks, frozen),
Qwen2.5-VL-3B projector (C = 37, H = 2048, conv fusion; trainable), 1024^2 latents (4096 image + 512 text tokens), random weights and
random teacher tensors.  HIP-event timing of every phase of x2i_amd.train.distill_step.   python tools/train_bench.py [B] [steps]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from x2i_amd.flux import FluxTransformer2DModel  # noqa: E402
from x2i_amd.proj import create_proj3_qwen3b  # noqa: E402
from x2i_amd.train import DistillBackward, ProjectorTrainer  # noqa: E402


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(pos[0]) if len(pos) > 0 else 1
    steps = int(pos[1]) if len(pos) > 1 else 3
    dev = "cuda"
    m = FluxTransformer2DModel(guidance_embeds=True, device=dev).init_random_(seed=1)
    pr = create_proj3_qwen3b(in_channels=37, use_t5=False, use_scale=False, use_cnn=True, device=dev).init_random_(2)
    tr = ProjectorTrainer(pr, lr=1e-5)
    chain = DistillBackward(m)
    g = torch.Generator(device=dev).manual_seed(3)
    St, Si, D = 512, 4096, 3072
    x = (torch.randn((B, 37, St, 2048), device=dev, generator=g) * 3).bfloat16()
    lat = torch.randn((B, Si, 64), device=dev, generator=g).bfloat16()
    ts = torch.full((B,), 0.6, device=dev)
    gd = torch.full((B,), 3.5, device=dev)
    from x2i_amd.pipeline import FluxPipeline
    img_ids = FluxPipeline._prepare_latent_image_ids(1, 64, 64, dev, torch.float32)
    txt_ids = torch.zeros((St, 3), device=dev)
    teacher = [(torch.randn((B, 19, Si, D), device=dev, generator=g)).bfloat16(), (torch.randn((B, 19, St, D), device=dev, generator=g)).bfloat16(),
               (torch.randn((B, 38, St + Si, D), device=dev, generator=g)).bfloat16()]
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    names = ("projector forward", "transformer forward (saving) + loss at 76 taps", "transformer backward (activation gradients)",
             "projector backward", "all-reduce + clip + AdamW")
    tot = [0.0] * 5
    for it in range(steps + 1):
        e = [ev() for _ in range(6)]
        e[0].record()
        pooled, prompt = tr.forward(x)
        e[1].record()
        st = chain.prepare_conditioning(prompt, pooled, txt_ids, img_ids, gd)
        _, loss = chain.forward_train(st, lat, ts, teacher=teacher)
        e[2].record()
        d_enc, d_pooled = chain.backward()
        e[3].record()
        tr.backward(d_enc, d_pooled)
        e[4].record()
        coef = tr.step()
        e[5].record()
        torch.cuda.synchronize()
        dt = [e[i].elapsed_time(e[i + 1]) for i in range(5)]
        print(f"step {it}: loss {float(loss):.4f}  grad norm {float(coef[1]):.4e}  " + "  ".join(f"{v:8.1f}" for v in dt) + f"  total {sum(dt):8.1f} ms"
              + ("  (first step: weight transposes, allocations)" if it == 0 else ""), flush=True)
        if it > 0:
            tot = [a + b for a, b in zip(tot, dt)]
    if "--graph" in sys.argv:
        from x2i_amd.train import GraphedDistillStep
        gs = GraphedDistillStep(tr, chain, txt_ids, img_ids, gd)
        for it in range(steps + 2):   # eager warm-up, capture, then replays
            e0, e1 = ev(), ev()
            e0.record()
            loss = gs(x, lat, ts, teacher)
            e1.record()
            torch.cuda.synchronize()
            print(f"graphed step {it}: loss {float(loss):.4f}  total {e0.elapsed_time(e1):8.1f} ms" + ("  (eager warm-up)" if it == 0 else "  (capture + replay)" if it == 1 else ""),
                  flush=True)
    print(f"B = {B}, 1024^2, 19 + 38 blocks; mean over {steps} steps:")
    for n, v in zip(names, tot):
        print(f"  {n:50s} {v / steps:9.1f} ms")
    print(f"  {'whole step':50s} {sum(tot) / steps:9.1f} ms   ({B * steps / (sum(tot) * 1e-3):.3f} samples/s)")
    print(f"  peak device memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
