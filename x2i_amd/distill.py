"""Attention-distillation capture (reference train/train_qwenvl.py:186-214,556-654).

What the reference's training step needs from the diffusion transformer, and what this build provides on the HIP path:

  * hooks on every block's `attn` module collecting the attention outputs of the teacher (frozen FLUX) and of the student
    (same frozen FLUX driven by the trainable projector's embeddings): `cast_hook_list` below is the reference's helper with the
    same name and list layout; the HIP transformer materialises those outputs only while hooks are registered
    (x2i_amd/flux.py `_AttnTap`, x2i_gated_residual_bf16) -- PROVIDED;
  * the per-block loss  KL( softmax(normalize(teacher) / 3) || softmax(normalize(student) / 3) ), `kd_attention_loss` below:
    plain torch on the captured tensors (it is a few reductions over tensors that already exist; not a hot-path kernel) -- PROVIDED;
  * the gradient of that loss with respect to the projector's parameters: it flows BACKWARDS THROUGH ALL 57 TRANSFORMER BLOCKS
    (train/train_qwenvl.py:626 `loss.backward()` with only proj_t5 trainable).  That chain, the loss kernel, the projector's
    backward, gradient clipping and AdamW live in x2i_amd/train.py (`DistillBackward`, `ProjectorTrainer`, `distill_step`) on the
    HIP path; `teacher_student_loss` below is the forward-only helper (loss value from two hooked forwards).
The gather / scatter of prompts and teacher tensors between teacher ranks and training ranks (core/pipeline/train_and_infer.py:31-122)
is x2i_amd/dist.py (`TeacherStudentGroups`, `send_to_infer_device`, `receive_from_infer_device`).
"""
import torch
import torch.nn.functional as F


def cast_hook_list(unet, lists):
    """train/train_qwenvl.py:206-214: lists[0] / lists[1] <- (image, text) attention outputs of the double blocks, lists[2] <- the
    single blocks' attention outputs.  Returns the hook handles (the reference drops them)."""
    lists.append([])
    lists.append([])
    lists.append([])
    handles = []

    def two(list0, list1):
        def hook(model, input, output):
            list0.append(output[0])
            list1.append(output[1])
        return hook

    def one(lst):
        def hook(model, input, output):
            lst.append(output)
        return hook

    for net in unet.transformer_blocks:
        handles.append(net.attn.register_forward_hook(two(lists[0], lists[1])))
    for net in unet.single_transformer_blocks:
        handles.append(net.attn.register_forward_hook(one(lists[2])))
    return handles


def normalize(logit):
    """train/train_qwenvl.py:58-61"""
    mean = logit.mean(dim=-1, keepdims=True)
    stdv = logit.std(dim=-1, keepdims=True)
    return (logit - mean) / (1e-7 + stdv)


def kd_attention_loss(teacher, student, temperature=3.0):
    """train/train_qwenvl.py:613-634.  teacher / student: three tensors [B, n_blocks, S, D] (torch.stack(list, dim=1)) or the three
    hook lists themselves.  Non-finite per-block terms are skipped, as in the reference."""
    loss = 0.0
    for t_all, s_all in zip(teacher, student):
        if isinstance(t_all, (list, tuple)):
            t_all = torch.stack(list(t_all), dim=1)
        if isinstance(s_all, (list, tuple)):
            s_all = torch.stack(list(s_all), dim=1)
        for i in range(t_all.shape[1]):
            term = F.kl_div(F.softmax(normalize(t_all[:, i].float()) / temperature, dim=-1).log(),
                            F.softmax(normalize(s_all[:, i].float()) / temperature, dim=-1), reduction="batchmean")
            if not (torch.isinf(term).any() or torch.isnan(term).any()):
                loss = loss + term
    return loss


@torch.no_grad()
def teacher_student_loss(transformer, teacher_inputs, student_inputs, temperature=3.0):
    """One forward of the (frozen) transformer on the teacher's conditioning and one on the student's (the projector's) with the
    reference's hooks attached; returns (loss, teacher_lists, student_lists).  Forward only (see the module docstring)."""
    out = []
    for kw in (teacher_inputs, student_inputs):
        lists = []
        handles = cast_hook_list(transformer, lists)
        try:
            transformer(**kw)
        finally:
            for h in handles:
                h.remove()
        out.append(lists)
    return kd_attention_loss(out[0], out[1], temperature), out[0], out[1]
