"""N2 hand-off contract against real HF code (tiny random Qwen2 on the CPU): the hook-filled [B, C, S, H] slab of ONE prefill
forward equals what the reference stacks from generate(..., output_hidden_states=True).hidden_states[0]."""
import pytest
import torch

transformers = pytest.importorskip("transformers")


def tiny_qwen2():
    from transformers import Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(0)
    cfg = Qwen2Config(vocab_size=120, hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=128)
    return Qwen2ForCausalLM(cfg).eval()


def test_slab_equals_generate_prompt_pass():
    from x2i_amd.handoff import HiddenStateSlab, find_decoder, prefill_hidden_states
    from x2i_amd.infer.harness import stack_hidden_states
    m = tiny_qwen2()
    ids = torch.randint(0, 120, (2, 9), generator=torch.Generator().manual_seed(1))
    mask = torch.ones_like(ids)
    out = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=4, output_hidden_states=True, return_dict_in_generate=True,
                     do_sample=False)
    want = stack_hidden_states(out.hidden_states)  # reference contract: torch.stack(hidden_states[0], dim=1)
    got = prefill_hidden_states(m, dtype=torch.float32, input_ids=ids, attention_mask=mask)
    assert got.shape == want.shape == (2, 4, 9, 32)
    assert torch.allclose(got, want, atol=1e-5, rtol=1e-5)
    # reusable slab: a second call with the same shapes writes into the same storage; hooks are gone afterwards
    slab = HiddenStateSlab(find_decoder(m), torch.float32)
    a = slab.prefill(m, input_ids=ids, attention_mask=mask)
    p = a.data_ptr()
    b = slab.prefill(m, input_ids=ids.flip(1), attention_mask=mask)
    assert b.data_ptr() == p and not torch.allclose(b, want)
    assert not any(len(l._forward_pre_hooks) for l in find_decoder(m).layers)


def test_slab_feeds_the_projector_layout():
    """C = n_layers + 1 and [B, C, S, H] is what Proj7Exp takes (utils/proj.py:62-72)."""
    from x2i_amd.handoff import prefill_hidden_states
    m = tiny_qwen2()
    ids = torch.randint(0, 120, (1, 5))
    x = prefill_hidden_states(m, input_ids=ids, attention_mask=torch.ones_like(ids))
    assert x.dtype == torch.bfloat16 and x.shape == (1, m.config.num_hidden_layers + 1, 5, 32) and x.is_contiguous()


def test_find_decoder_rejects_models_without_a_stack():
    from x2i_amd.handoff import find_decoder
    with pytest.raises(RuntimeError):
        find_decoder(torch.nn.Linear(2, 2))


class _MiniCPMLike(torch.nn.Module):
    """Shape of MiniCPM-o's wrapper as the reference drives it (infer/inference_minicpm.py:174-176): `.llm` is the HF causal LM,
    generate(**inputs, tokenizer=, max_new_tokens=1, decode_text=False) builds inputs_embeds (vision / audio features spliced in),
    runs the LLM's generate and -- in the reference's patched copy -- returns an object with `.hidden_states`."""

    def __init__(self):
        super().__init__()
        self.llm = tiny_qwen2()
        self.extra = torch.nn.Parameter(torch.randn(1, 1, 32) * 0.1)  # stands in for the spliced multimodal features

    def generate(self, input_ids=None, attention_mask=None, tokenizer=None, max_new_tokens=1, decode_text=False, **kw):
        emb = self.llm.get_input_embeddings()(input_ids) + self.extra
        return self.llm.generate(inputs_embeds=emb, attention_mask=attention_mask, max_new_tokens=max_new_tokens, do_sample=False,
                                 output_hidden_states=True, return_dict_in_generate=True)


class _InternVLLike(torch.nn.Module):
    """InternVL chat model: `.language_model` is the HF causal LM; the STOCK generate() returns token ids after several decoder
    passes, the reference's patched one returns the prompt pass' hidden states (modeling_internvl_chat.py:314-363)."""

    def __init__(self):
        super().__init__()
        self.language_model = tiny_qwen2()

    def generate(self, pixel_values=None, input_ids=None, attention_mask=None, max_new_tokens=3, **kw):
        emb = self.language_model.get_input_embeddings()(input_ids)
        if pixel_values is not None:
            emb = emb + pixel_values.mean() * 0.01
        return self.language_model.generate(inputs_embeds=emb, attention_mask=attention_mask, max_new_tokens=max_new_tokens, do_sample=False)

    def reference_patched_generate(self, pixel_values, input_ids, attention_mask):
        emb = self.language_model.get_input_embeddings()(input_ids)
        if pixel_values is not None:
            emb = emb + pixel_values.mean() * 0.01
        return self.language_model(inputs_embeds=emb, attention_mask=attention_mask, output_hidden_states=True).hidden_states


def test_minicpm_conditioner_slab_equals_reference_stack():
    """Row N2 for MiniCPM: hooks during the model's own generate(max_new_tokens=1) == torch.stack(hidden_states[0], dim=1)."""
    from x2i_amd.infer.harness import stack_hidden_states
    from x2i_amd.infer.inference_minicpm import MiniCPMConditioner
    m = _MiniCPMLike().eval()
    ids = torch.randint(0, 120, (1, 11), generator=torch.Generator().manual_seed(2))
    inputs = dict(input_ids=ids, attention_mask=torch.ones_like(ids))
    fast = MiniCPMConditioner(None, "cpu", prefill_only=True, model=m)
    fast.slab.dtype = torch.float32
    ref = MiniCPMConditioner(None, "cpu", prefill_only=False, model=m)
    got, want = fast.hidden_states(inputs), ref.hidden_states(inputs)
    assert want.shape == (1, 4, 11, 32) and torch.allclose(got, want, atol=1e-5, rtol=1e-5)
    assert torch.allclose(want, stack_hidden_states(m.generate(**inputs).hidden_states))


def test_internvl_conditioner_slab_works_with_stock_generate():
    """Row N2 for InternVL: the slab keeps the FIRST decoder pass of a multi-token stock generate(); equal to what the reference's
    patched one-forward generate() returns (stacked)."""
    from x2i_amd.infer.inference_internvl import InternVLConditioner
    m = _InternVLLike().eval()
    ids = torch.randint(0, 120, (1, 13), generator=torch.Generator().manual_seed(3))
    mask = torch.ones_like(ids)
    pix = torch.randn(1, 3, 8, 8)
    c = InternVLConditioner(None, "cpu", prefill_only=True, model=m)
    c.slab.dtype = torch.float32
    got = c.hidden_states(pix, ids, mask)
    want = torch.stack(tuple(m.reference_patched_generate(pix, ids, mask)), dim=1)
    assert got.shape == want.shape == (1, 4, 13, 32) and torch.allclose(got, want, atol=1e-5, rtol=1e-5)
    # a multi-token generate(): the later decode passes run and are ignored, the slab still holds the prompt pass
    again = c.slab.capture(lambda: m.generate(pixel_values=pix, input_ids=ids, attention_mask=mask, max_new_tokens=3))
    assert c.slab._extra > 0 and torch.allclose(again, want, atol=1e-5, rtol=1e-5)
