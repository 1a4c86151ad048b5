"""Host-side logic of the sampling harness (Row H) that needs no GPU and no checkpoints."""
import torch

from x2i_amd.infer import harness as H


def test_cli_flags_match_reference_surface():
    a = H.build_parser("qwenvl").parse_args([])
    assert (a.qwen_size, a.num_steps, a.num_gen_imgs, a.task, a.use_answer) == ("7b", 4, 1, "all", False)
    assert H.build_parser("qwenvl").parse_args(["--use_answer", "x"]).use_answer is True  # type=bool quirk preserved
    assert H.build_parser("internvl").parse_args(["--internvl_size", "1b"]).internvl_size == "1b"
    assert H.build_parser("minicpm").parse_args([]).minicpm_path == "openbmb/MiniCPM-o-2_6"
    for t in ("text2image", "image2image", "imagetext2image", "video2image", "audio2image", "x2image"):
        assert H.build_parser("minicpm").parse_args(["--task", t]).task == t


def test_hidden_state_stacking_contract():
    B, C, S, Hd, T = 1, 5, 7, 16, 3
    prompt = tuple(torch.randn(B, S, Hd) for _ in range(C))
    gen = [tuple(torch.randn(B, 1, Hd) for _ in range(C)) for _ in range(T)]
    hs = (prompt,) + tuple(gen)
    x = H.stack_hidden_states(hs)
    assert x.shape == (B, C, S, Hd)
    assert torch.equal(x, torch.cat(prompt).unsqueeze(0))  # the reference's B=1 form (infer/inference_qwenvl.py:123)
    xa = H.stack_hidden_states(hs, use_answer=True)
    ref = torch.cat([torch.cat(g) for g in gen], dim=1).unsqueeze(0)  # :125-129
    assert xa.shape == (B, C, T, Hd) and torch.equal(xa, ref)
    Bb = 3
    promptb = tuple(torch.randn(Bb, S, Hd) for _ in range(C))
    assert torch.equal(H.stack_hidden_states((promptb,)), torch.stack(promptb, dim=1))  # inference_minicpm.py:117


def test_projector_table_and_prefix_strip():
    assert {k: v[1] for k, v in H.PROJECTORS.items()} == dict(qwen3b=37, qwen7b=29, internvl1b=25, internvl4b=37, minicpm=29)
    assert H.PROJECTORS["internvl1b"][2]["use_scale"] is True and H.PROJECTORS["qwen7b"][2]["use_cnn"] is True
    sd = {"module.mlp.fc.1.weight": 1, "conv.bias": 2}
    assert H.strip_module_prefix(sd) == {"mlp.fc.1.weight": 1, "conv.bias": 2}
    for kind, (make, C, kw) in H.PROJECTORS.items():
        p = make(in_channels=C, device="meta", **kw)
        assert p.mlp.layernorm.weight.shape[0] == H.HIDDEN[kind]


def test_real_prompt_batching_groups_by_text_length():
    """generate_jobs packs prompts per text length (FLUX's joint attention has no mask, so padding is not an option)."""
    conds = [(torch.zeros(1, 768), torch.zeros(1, s, 4096)) for s in (512, 77, 512, 300, 77)]
    assert H.Harness.group_by_length(conds) == [[0, 2], [1, 4], [3]]
