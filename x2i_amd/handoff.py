"""MLLM -> projector hand-off.

The reference gets its conditioning tensor by running `generate(max_new_tokens=128, output_hidden_states=True)` and then
`torch.cat/stack`-ing the per-layer tuple of the PROMPT pass into [B, C, S, H] (infer/inference_qwenvl.py:121-132,176-179;
inference_minicpm.py:116-118,174-177): 127 decode steps whose states are thrown away unless --use_answer, plus a 78-106 MB
copy per sample.  `HiddenStateSlab` produces the same tensor from ONE prefill forward: hooks on the decoder stack write every
layer's hidden state straight into a preallocated [B, C, S, H] buffer (C = n_layers + 1, the HF `hidden_states` convention:
entry i < n_layers is the INPUT of layer i, the last entry is the output of the final norm), which is the projector's input.
Plain PyTorch on purpose: the MLLM is not on the accelerated path, only the layout contract with it is.
"""
import torch
import torch.nn as nn


def find_decoder(model):
    """The text decoder stack inside an HF causal / conditional-generation model: the module owning `.layers` (ModuleList)
    and `.norm`.  Qwen2.5-VL, MiniCPM-o and InternVL all wrap a Qwen2/LLaMA-style decoder of this shape."""
    best = None
    for m in model.modules():
        layers = getattr(m, "layers", None)
        if isinstance(layers, nn.ModuleList) and hasattr(m, "norm") and hasattr(m, "embed_tokens"):
            if best is None or len(layers) > len(best.layers):
                best = m
    if best is None:
        raise RuntimeError("handoff: no decoder stack (layers + norm + embed_tokens) found in %s" % type(model).__name__)
    return best


class HiddenStateSlab:
    def __init__(self, decoder, dtype=torch.bfloat16):
        self.decoder, self.dtype = decoder, dtype
        self.n_layers = len(decoder.layers)
        self.C = self.n_layers + 1
        self.slab = None
        self._handles = []
        self._seen = 0
        self._extra = 0

    # ---- hooks
    def _store(self, idx, h):
        if self._seen >= self.C:  # a later decoder pass (a decode step of generate()): the conditioning is the PROMPT pass only
            self._extra += 1
            return
        if self.slab is None or self.slab.shape[0] != h.shape[0] or self.slab.shape[2] != h.shape[1] or self.slab.device != h.device:
            self.slab = torch.empty((h.shape[0], self.C, h.shape[1], h.shape[2]), device=h.device, dtype=self.dtype)
        self.slab[:, idx].copy_(h)
        self._seen += 1

    def attach(self):
        self.detach()
        for i, layer in enumerate(self.decoder.layers):
            def pre(mod, args, kwargs, i=i):
                h = args[0] if args else kwargs["hidden_states"]
                self._store(i, h)
            self._handles.append(layer.register_forward_pre_hook(pre, with_kwargs=True))
        self._handles.append(self.decoder.norm.register_forward_hook(lambda mod, args, out: self._store(self.n_layers, out)))
        return self

    def detach(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def __enter__(self):
        return self.attach()

    def __exit__(self, *exc):
        self.detach()

    # ---- one prefill forward -> [B, C, S, H]
    @torch.no_grad()
    def prefill(self, model, **inputs):
        return self.capture(lambda: model(**inputs, use_cache=False))

    @torch.no_grad()
    def capture(self, run):
        """Drive the MLLM with any callable -- a plain forward, or the model's own multimodal `generate(max_new_tokens=1)` /
        `chat()` wrapper that builds inputs_embeds from pixels / audio first (MiniCPM-o, InternVL) -- and keep the hidden states
        of the FIRST decoder pass, i.e. the prompt pass the reference stacks (infer/inference_minicpm.py:116-118,174-177;
        infer/inference_internvl.py:159-188).  Later passes (decode steps) are ignored, so this works with the stock HF
        remote code and does not need the reference's patched `generate()` that returns hidden states."""
        self._seen = self._extra = 0
        with self:
            run()
        if self._seen != self.C:
            raise RuntimeError("handoff: captured %d hidden states, expected %d (did the decoder stack run?)" % (self._seen, self.C))
        return self.slab


def prefill_hidden_states(model, dtype=torch.bfloat16, **inputs):
    """Convenience wrapper: [B, C, S, H] conditioning tensor of `inputs` from one forward of `model`."""
    return HiddenStateSlab(find_decoder(model), dtype).prefill(model, **inputs)
