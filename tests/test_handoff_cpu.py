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
