#!/usr/bin/env python3
"""MiniCPM-o-2.6 (image / video / audio / text) -> X2I sampling on the HIP path.  Counterpart of infer/inference_minicpm.py:
unpadded inputs, generate(max_new_tokens=1, decode_text=False), torch.stack(hidden_states[0], dim=1) (:116-118,160-177)."""
import torch

from .harness import Harness, SyntheticConditioner, asset, build_parser, stack_hidden_states

MAX_NUM_FRAMES = 64


def encode_video(path):
    """1 frame per second, uniformly thinned to MAX_NUM_FRAMES (inference_minicpm.py:120-135)."""
    from decord import VideoReader, cpu
    from PIL import Image
    vr = VideoReader(path, ctx=cpu(0))
    idx = list(range(0, len(vr), max(1, round(vr.get_avg_fps()))))
    if len(idx) > MAX_NUM_FRAMES:
        gap = len(idx) / MAX_NUM_FRAMES
        idx = [idx[int(i * gap + gap / 2)] for i in range(MAX_NUM_FRAMES)]
    return [Image.fromarray(f.astype("uint8")) for f in vr.get_batch(idx).asnumpy()]


class MiniCPMConditioner:
    """prefill_only (default): the [1, C, S, H] conditioning tensor is written by hooks on the LLM decoder during the ONE prompt
    pass of generate(max_new_tokens=1) (x2i_amd/handoff.py, row N2) -- no hidden-state tuple, no torch.stack copy, and no
    dependence on the reference's patched generate() returning `.hidden_states`.  full_generate: the reference's form."""

    def __init__(self, path, device, prefill_only=True, model=None, tokenizer=None, processor=None):
        if model is None:
            from transformers import AutoModel, AutoProcessor, AutoTokenizer
            model = AutoModel.from_pretrained(path, trust_remote_code=True, torch_dtype=torch.bfloat16).eval().to(device)
            tokenizer = AutoTokenizer.from_pretrained(path, trust_remote_code=True)
            processor = AutoProcessor.from_pretrained(path, trust_remote_code=True)
        self.model, self.tokenizer, self.processor = model, tokenizer, processor
        self.device = device
        self.slab = None
        if prefill_only:
            from ..handoff import HiddenStateSlab, find_decoder
            self.slab = HiddenStateSlab(find_decoder(self.model))

    def hidden_states(self, inputs):
        """processor outputs -> [B, C, S, H]"""
        run = lambda: self.model.generate(**inputs, tokenizer=self.tokenizer, max_new_tokens=1, decode_text=False)  # noqa: E731
        if self.slab is not None:
            return self.slab.capture(run)
        return stack_hidden_states(run().hidden_states)

    @torch.no_grad()
    def __call__(self, videos=None, images=None, audios=None, text_prompt=None):
        from PIL import Image
        text, image_list, audio_list = "", [], []
        for p in images or []:
            image_list.append(Image.open(p).convert("RGB"))
            text += "(<image>./</image>)\n"
        for v in videos or []:
            frames = encode_video(v)
            text += "(<image>./</image>)\n" * len(frames)
            image_list.extend(frames)
        for a in audios or []:
            import librosa
            wav, _ = librosa.load(a, sr=16000, mono=True)
            text += "(<audio>./</audio>)\n"
            audio_list.append(wav)
        text += text_prompt or ""
        prompt = self.processor.tokenizer.apply_chat_template([{"role": "user", "content": text}], tokenize=False,
                                                              add_generation_prompt=True)
        inputs = self.processor(text=[prompt], images=[image_list] if image_list else None,
                                audios=[audio_list] if audio_list else None, max_slice_nums=1, use_image_id=False,
                                chunk_input=True, return_tensors="pt", max_length=32768, sampling_rate=16000,
                                add_special_tokens=True).to(self.device)
        inputs.pop("image_sizes")
        return self.hidden_states(inputs)


def tasks(args):
    img = lambda n: asset(args, "image", n)
    return {
        "text2image": [dict(filename="elephant", text_prompt="A majestic elephant in a sun-drenched savannah.")],
        "image2image": [dict(filename="sea_moon", images=[img("sea_moon.jpg")])],
        "imagetext2image": [dict(filename="man_smile", images=[img("man.jpg")], text_prompt="Make the person in the picture smile")],
        "video2image": [dict(filename="Skiing", videos=[asset(args, "video", "Skiing.mp4")])],
        "audio2image": [dict(filename="audio", audios=[asset(args, "audio", "thunder.wav")])],
    }


def main(argv=None):
    args = build_parser("minicpm").parse_args(argv)
    device = "cuda:%d" % int(__import__("os").environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    cond = SyntheticConditioner("minicpm", device) if args.synthetic else MiniCPMConditioner(args.minicpm_path, device,
                                                                                              prefill_only=not args.full_generate)
    Harness(args, "minicpm", cond, device).run_tasks(tasks(args))


if __name__ == "__main__":
    main()
