#!/usr/bin/env python3
"""Interactive multi-turn Qwen2.5-VL -> image loop on the HIP path.  Counterpart of infer/inference_multi_turn.py: the
chat history grows each turn, the model answers with max_new_tokens=64, and the conditioning is the prompt-pass hidden
states concatenated along S with the generated-token hidden states (:132-141); 4 steps, 1024x1024, manual_seed(0)."""
import time

import torch

from .harness import Harness, SyntheticConditioner, build_parser, stack_hidden_states


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


def synthetic_turns(args, kind, device):
    """--synthetic: no checkpoints; the conversation is a list of text lengths (--turn_lengths, default: a history that grows by 37 tokens
    per turn, then the first two lengths again) and every turn prints its sampling latency -- a NEW length runs the eager launch sequence
    once (FluxPipeline's graph policy: no discarded warm-up pass), a length that comes back is captured and from then on replayed."""
    lengths = [int(x) for x in args.turn_lengths.split(",")] if args.turn_lengths else [96 + 37 * t for t in range(4)] + [96, 133, 96, 133]
    cond = SyntheticConditioner(kind, device)
    h = Harness(args, kind, cond, device)
    for turn, S in enumerate(lengths):
        cond.seq_len = S
        pooled, embeds = h.embeds(text_prompt="turn %d" % turn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.generate(pooled, embeds, "multi_turn", "turn_%d" % turn, seed=0)
        torch.cuda.synchronize()
        st = h.pipeline.graph_stats
        print("turn %d: S_txt %4d  sampling %8.1f ms   (pipeline so far: %d eager, %d captures, %d replays)"
              % (turn, S, (time.perf_counter() - t0) * 1e3, st["eager"], st["captures"], st["replays"]), flush=True)


def main(argv=None):
    ap = build_parser("qwenvl")
    ap.add_argument("--turn_lengths", type=str, default=None, help="with --synthetic: comma-separated text lengths of the turns")
    args = ap.parse_args(argv)
    kind = "qwen" + args.qwen_size
    device = "cuda:0"
    torch.cuda.set_device(device)
    if args.synthetic:
        return synthetic_turns(args, kind, device)
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
