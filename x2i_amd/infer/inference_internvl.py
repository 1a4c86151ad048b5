#!/usr/bin/env python3
"""InternVL2.5 (1B / 4B) -> X2I sampling on the HIP path.  Counterpart of infer/inference_internvl.py: tokens padded to
512 (:126), the (locally modified) chat model's generate() is ONE forward that returns every layer's hidden states
(model_internvl/internvl/modeling_internvl_chat.py:314-363), default seed 1 (:192), tasks text2image / image2image /
imagetext2image only (:288-294)."""
import torch

from .harness import Harness, SyntheticConditioner, asset, build_parser


class InternVLConditioner:
    """prefill_only (default): hooks on the language model's decoder stack fill the [1, C, 512, H] tensor during the first
    decoder pass of the chat model's generate() (x2i_amd/handoff.py, row N2).  That works with the STOCK InternVL2.5 remote code
    (whose generate() returns token ids); full_generate expects the reference's locally modified modeling_internvl_chat.py, whose
    generate() is one forward returning every layer's hidden states (model_internvl/internvl/modeling_internvl_chat.py:314-363)."""

    def __init__(self, path, device, prefill_only=True, model=None, tokenizer=None):
        if model is None:
            from transformers import AutoModel, AutoTokenizer
            model = AutoModel.from_pretrained(path, trust_remote_code=True, torch_dtype=torch.bfloat16).eval().to(device)
            tokenizer = AutoTokenizer.from_pretrained(path, trust_remote_code=True, use_fast=False)
        self.model, self.tokenizer = model, tokenizer
        self.device = device
        self.slab = None
        if prefill_only:
            from ..handoff import HiddenStateSlab, find_decoder
            self.slab = HiddenStateSlab(find_decoder(self.model))

    def hidden_states(self, pixel_values, input_ids, attention_mask):
        if self.slab is not None:
            return self.slab.capture(lambda: self.model.generate(pixel_values=pixel_values, input_ids=input_ids,
                                                                 attention_mask=attention_mask, max_new_tokens=1))
        hs = self.model.generate(pixel_values=pixel_values, input_ids=input_ids, attention_mask=attention_mask)
        return torch.stack(tuple(hs), dim=1)

    @torch.no_grad()
    def __call__(self, videos=None, images=None, audios=None, text_prompt=None):
        pixel_values = None
        if images:
            from PIL import Image
            import numpy as np
            tiles = []
            for p in images:  # 128x128 resize then a single 448x448 ImageNet-normalised tile (inference_internvl.py:171)
                im = Image.open(p).convert("RGB").resize((128, 128)).resize((448, 448))
                a = torch.from_numpy(np.asarray(im)).float().div(255).permute(2, 0, 1)
                mean, std = torch.tensor([0.485, 0.456, 0.406])[:, None, None], torch.tensor([0.229, 0.224, 0.225])[:, None, None]
                tiles.append((a - mean) / std)
            pixel_values = torch.stack(tiles).to(self.device, torch.bfloat16)
        q = ("<image>\n" * len(images or [])) + (text_prompt or "")
        tok = self.tokenizer(q, padding="max_length", max_length=512, truncation=True, return_tensors="pt").to(self.device)
        return self.hidden_states(pixel_values, tok.input_ids, tok.attention_mask)


def tasks(args):
    img = lambda n: asset(args, "image", n)
    return {
        "text2image": [dict(filename="elephant", text_prompt="A majestic elephant in a sun-drenched savannah.")],
        "image2image": [dict(filename="sea_moon", images=[img("sea_moon.jpg")])],
        "imagetext2image": [dict(filename="hutong_car", images=[img("hutong.jpg")], text_prompt="Add a car in the picture")],
    }


def main(argv=None):
    p = build_parser("internvl")
    p.set_defaults(seed=1)
    args = p.parse_args(argv)
    kind = "internvl" + args.internvl_size
    device = "cuda:%d" % int(__import__("os").environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    cond = SyntheticConditioner(kind, device) if args.synthetic else InternVLConditioner(args.internvl_path, device,
                                                                                         prefill_only=not args.full_generate)
    Harness(args, kind, cond, device).run_tasks(tasks(args))


if __name__ == "__main__":
    main()
