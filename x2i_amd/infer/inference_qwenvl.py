#!/usr/bin/env python3
"""Qwen2.5-VL (3B / 7B) -> X2I sampling on the HIP path.  Counterpart of infer/inference_qwenvl.py.

  python -m x2i_amd.infer.inference_qwenvl --qwen_size 7b --flux_path <dir> --proj_path <bin> --task text2image
  torchrun --nproc-per-node 8 --master-addr 127.0.0.1 -m x2i_amd.infer.inference_qwenvl --batch 32 ...
  python -m x2i_amd.infer.inference_qwenvl --synthetic --task text2image        # no checkpoints needed
"""
import torch

from .harness import Harness, SyntheticConditioner, asset, build_parser, stack_hidden_states


class QwenVLConditioner:
    """Qwen2_5_VLForConditionalGeneration prompt pass; inputs padded to 512 tokens, images resized to 128x128, videos at
    1 fps / 128*128 pixels, generate(max_new_tokens=128, output_hidden_states=True) -- infer/inference_qwenvl.py:136-180."""

    def __init__(self, path, device, use_answer=False, prefill_only=True):
        from transformers import AutoProcessor, Qwen2_5_VLForConditionalGeneration
        self.model = Qwen2_5_VLForConditionalGeneration.from_pretrained(path, torch_dtype=torch.bfloat16).eval().to(device)
        self.processor = AutoProcessor.from_pretrained(path)
        self.device, self.use_answer = device, use_answer
        # the prompt-pass states are all the reference keeps unless --use_answer: take them from ONE forward, written by
        # hooks straight into the [B,C,S,H] buffer (x2i_amd/handoff.py), instead of a 128-token generate() + torch.cat
        self.slab = None
        if prefill_only and not use_answer:
            from ..handoff import HiddenStateSlab, find_decoder
            self.slab = HiddenStateSlab(find_decoder(self.model))

    @torch.no_grad()
    def __call__(self, videos=None, images=None, audios=None, text_prompt=None):
        from PIL import Image
        content, image_list, video_inputs = [], [], None
        for path in images or []:
            im = Image.open(path).convert("RGB").resize(size=(128, 128))
            content.append({"type": "image", "image": im})
            image_list.append(im)
        if videos:
            assert len(videos) == 1
            from qwen_vl_utils import process_vision_info
            content.append({"type": "video", "video": videos[0], "max_pixels": 128 * 128, "fps": 1.0})
            _, video_inputs = process_vision_info([{"role": "user", "content": content}])
        if text_prompt is not None:
            content.append({"type": "text", "text": text_prompt})
        message = [{"role": "user", "content": content}]
        prompt = self.processor.apply_chat_template(message, tokenize=False, add_generation_prompt=True)
        inputs = self.processor(text=[prompt], images=image_list or None, videos=video_inputs, padding="max_length",
                                max_length=512, truncation=True, return_tensors="pt").to(self.device)
        if self.slab is not None:
            return self.slab.prefill(self.model, **inputs)
        out = self.model.generate(**inputs, max_new_tokens=128, output_hidden_states=True, return_dict_in_generate=True)
        return stack_hidden_states(out["hidden_states"], use_answer=self.use_answer)


def tasks(args):
    img = lambda n: asset(args, "image", n)
    vid = lambda n: asset(args, "video", n)
    return {
        "text2image": [dict(filename="elephant_%s" % k, text_prompt=p) for k, p in (
            ("EN", "A majestic elephant in a sun-drenched savannah, impressionistic style, low camera angle."),
            ("ZH", "一只雄伟的大象站在阳光普照的草原上，印象派风格，低机位。"),
            ("DE", "Ein majestätischer Elefant in einer sonnenüberfluteten Savanne, impressionistischer Stil."),
            ("FR", "Un éléphant majestueux dans une savane baignée de soleil, style impressionniste."),
            ("JA", "日差しに照らされたサバンナに立つ荘厳な象、印象派のスタイル。"),
            ("VI", "Một con voi uy nghi trên thảo nguyên đầy nắng, phong cách ấn tượng."))],
        "image2image": [dict(filename="sea_moon", images=[img("sea_moon.jpg")]),
                        dict(filename="dog_hat", images=[img("dog.jpg"), img("hat.jpg")])],
        "imagetext2image": [dict(filename="yarn_ball_panda", images=[img("yarn_ball.jpg")],
                                 text_prompt="Refer to the image style and generate a cute giant panda"),
                            dict(filename="hutong_car", images=[img("hutong.jpg")], text_prompt="Add a car in the picture")],
        "video2image": [dict(filename="particle_collision", videos=[vid("particle_collision.mp4")]),
                        dict(filename="Skiing", videos=[vid("Skiing.mp4")])],
        "x2image": [dict(filename="Shuimohua", images=[img("Shuimohua.jpg")]),
                    dict(filename="particle_collision_x", videos=[vid("particle_collision.mp4")])],
    }


def main(argv=None):
    args = build_parser("qwenvl").parse_args(argv)
    kind = "qwen" + args.qwen_size
    device = "cuda:%d" % int(__import__("os").environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    cond = SyntheticConditioner(kind, device) if args.synthetic else QwenVLConditioner(args.qwen_path, device, args.use_answer, prefill_only=not args.full_generate)
    Harness(args, kind, cond, device).run_tasks(tasks(args))


if __name__ == "__main__":
    main()
