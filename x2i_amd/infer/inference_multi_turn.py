#!/usr/bin/env python3
"""Interactive multi-turn Qwen2.5-VL -> image loop on the HIP path.  Counterpart of infer/inference_multi_turn.py: the
chat history grows each turn, the model answers with max_new_tokens=64, and the conditioning is the prompt-pass hidden
states concatenated along S with the generated-token hidden states (:132-141); 4 steps, 1024x1024, manual_seed(0)."""
import torch

from .harness import Harness, build_parser, stack_hidden_states


class MultiTurnConditioner:
    def __init__(self, path, device):
        from transformers import AutoProcessor, Qwen2_5_VLForConditionalGeneration
        self.model = Qwen2_5_VLForConditionalGeneration.from_pretrained(path, torch_dtype=torch.bfloat16).eval().to(device)
        self.processor = AutoProcessor.from_pretrained(path)
        self.device = device
        self.history = []

    @torch.no_grad()
    def __call__(self, images=None, text_prompt=None, **_):
        from PIL import Image
        content, ims = [], []
        for p in images or []:
            im = Image.open(p).convert("RGB").resize((256, 256))  # :92
            content.append({"type": "image", "image": im})
            ims.append(im)
        content.append({"type": "text", "text": text_prompt})
        self.history.append({"role": "user", "content": content})
        prompt = self.processor.apply_chat_template(self.history, tokenize=False, add_generation_prompt=True)
        all_ims = [c["image"] for m in self.history if m["role"] == "user" for c in m["content"] if c["type"] == "image"]
        inputs = self.processor(text=[prompt], images=all_ims or None, return_tensors="pt").to(self.device)
        out = self.model.generate(**inputs, max_new_tokens=64, output_hidden_states=True, return_dict_in_generate=True)
        answer = self.processor.batch_decode(out.sequences[:, inputs.input_ids.shape[1]:], skip_special_tokens=True)[0]
        self.history.append({"role": "assistant", "content": [{"type": "text", "text": answer}]})
        prompt_hs = stack_hidden_states(out.hidden_states)
        if len(out.hidden_states) > 1:
            return torch.cat([prompt_hs, stack_hidden_states(out.hidden_states, use_answer=True)], dim=2), answer
        return prompt_hs, answer


def main(argv=None):
    args = build_parser("qwenvl").parse_args(argv)
    kind = "qwen" + args.qwen_size
    device = "cuda:0"
    torch.cuda.set_device(device)
    cond = MultiTurnConditioner(args.qwen_path, device)
    h = Harness(args, kind, lambda **kw: cond(**kw)[0], device)
    turn = 0
    while True:
        try:
            line = input("user (text [| image path ...], empty to quit)> ").strip()
        except EOFError:
            break
        if not line:
            break
        parts = [s.strip() for s in line.split("|")]
        pooled, embeds = h.embeds(text_prompt=parts[0], images=parts[1:] or None)
        h.generate(pooled, embeds, "multi_turn", "turn_%d" % turn, seed=0)
        print("assistant>", cond.history[-1]["content"][0]["text"])
        turn += 1


if __name__ == "__main__":
    main()
