#!/usr/bin/env python3
"""Tools-only same-node comparator (never imported by the product): the restated reference transformer
(`oracle.flux.flux_forward`: plain torch ops, so on `cuda` in bf16 it launches hipBLASLt GEMMs + the vendor SDPA + eager elementwise
kernels -- what infer/inference_qwenvl.py:188-207 would run on this GPU through stock PyTorch-ROCm) beside
`x2i_amd.FluxTransformer2DModel` on the SAME weights and inputs: B = 4, 1024^2 (512 text + 4096 image tokens), all 19 + 38 blocks,
one denoise step each, interleaved rounds.

    python tools/eager_gpu_baseline.py [B] [--blocks 19 38]      # prints one JSON line
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flux as OF  # noqa: E402  (tools-only: the comparator IS the restated reference)
from oracle import sampler as OS  # noqa: E402

DEV = "cuda"


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
    nl, ns = 19, 38
    if "--blocks" in sys.argv:
        i = sys.argv.index("--blocks")
        nl, ns = int(sys.argv[i + 1]), int(sys.argv[i + 2])
    from x2i_amd.flux import FluxTransformer2DModel
    m = FluxTransformer2DModel(num_layers=nl, num_single_layers=ns, guidance_embeds=False, device=DEV).init_random_(seed=9, std=0.02)
    sd = {k: v.detach() for k, v in m.state_dict().items()}  # bf16, on the device: the eager path reads the same tensors
    cfg = dict(OF.DEFAULT_CFG, num_layers=nl, num_single_layers=ns, guidance_embeds=False)
    g = torch.Generator(device=DEV).manual_seed(4)
    hidden = torch.randn((B, 4096, 64), device=DEV, generator=g).bfloat16()
    enc = torch.randn((B, 512, 4096), device=DEV, generator=g).bfloat16()
    pooled = torch.randn((B, 768), device=DEV, generator=g).bfloat16()
    t = torch.full((B,), 0.75, device=DEV).bfloat16()
    img_ids, txt_ids = OS.prepare_latent_image_ids(64, 64).to(DEV), torch.zeros(512, 3, device=DEV)

    def ours():
        return m(hidden_states=hidden, encoder_hidden_states=enc, pooled_projections=pooled, timestep=t, img_ids=img_ids,
                 txt_ids=txt_ids, guidance=None, return_dict=False)[0]

    # torch 2.10.0+rocm7.0 on this image faults ("Write access to a read-only page") inside the eager double-stream block at B = 4
    # (also with TORCH_BLAS_PREFER_HIPBLASLT=0; B <= 2 and the single-stream blocks are fine; profiles/r03g_*): the eager path is
    # therefore run in chunks of at most 2 samples, back to back -- its per-sample cost does not depend on the chunking
    chunk = 2 if B > 2 and "--eager-whole-batch" not in sys.argv else B

    def eager():
        with torch.no_grad():
            return torch.cat([OF.flux_forward(sd, cfg, hidden[i:i + chunk], enc[i:i + chunk], pooled[i:i + chunk], t[i:i + chunk], img_ids, txt_ids)
                              for i in range(0, B, chunk)], 0)

    def timed(f, iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            f()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters

    def stage(msg):
        torch.cuda.synchronize()
        print("[eager_gpu_baseline] " + msg, file=sys.stderr, flush=True)

    stage("model + inputs ready")
    o = ours()
    stage("x2i forward done")
    r = eager()
    stage("eager forward done")
    err = ((o.float() - r.float()).norm() / r.float().norm()).item()
    for f in (ours, eager):
        for _ in range(2):
            f()
    torch.cuda.synchronize()
    to, te = [], []
    for _ in range(4):
        to.append(timed(ours, 4))
        te.append(timed(eager, 4))
    flop = B * 74.38e12 * (nl * 1.0 / 19 * 0.5 + ns * 1.0 / 38 * 0.5) if (nl, ns) != (19, 38) else B * 74.38e12
    med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
    print(json.dumps({"workload": f"one denoise step, B={B}, 1024^2, {nl}+{ns} blocks, bf16, same weights/inputs",
                      "x2i_ms": round(med(to), 2), "eager_torch_rocm_ms": round(med(te), 2), "speedup": round(med(te) / med(to), 3),
                      "x2i_TFLOPs": round(flop / med(to) / 1e9, 1), "eager_TFLOPs": round(flop / med(te) / 1e9, 1),
                      "rel_l2_x2i_vs_eager_bf16": err, "eager_chunk": chunk, "torch": torch.__version__, "device": torch.cuda.get_device_name(0)}))


if __name__ == "__main__":
    main()
