"""Attention-distillation training harness: the counterpart of the reference's train/train_qwenvl.py main loop (:366-654) on the HIP path.

    python -m x2i_amd.train_distill --synthetic --max_train_steps 10 --batch_size 1 --output_dir out
    torchrun --nproc-per-node N -m x2i_amd.train_distill ...       (trainers form the data-parallel group; --local_infer_world_size K
                                                                   dedicates K ranks per node to the teacher, as the reference does)

Same argument names and defaults as the reference for what exists here (optimizer, clipping, accumulation, schedule, checkpointing);
checkpoints are the projector's state dict under <output_dir>/<global_step>/diffusion_pytorch_model.bin (:643-648), the format
x2i_amd.checkpoints / the inference scripts load.  What the reference takes from its data module (MLLM hidden states, latents, timesteps)
and from its teacher ranks (the three stacked attention tensors) arrives here as a batch dict with the reference's keys
(`text_embeddings`, `latents`, `timestep`, `KD_teacher_tensor0/1/2`, :570-578): `--synthetic` draws such batches at the reference's shapes
(no datasets or checkpoints are available offline), `--batch_files` reads them from torch-saved files.
"""
import argparse
import math
import os

import torch

from . import dist as xdist
from . import ops
from .flux import FluxTransformer2DModel
from .pipeline import FluxPipeline
from .proj import Proj7Exp, create_proj3_qwen3b, create_proj3_qwen7b
from .train import DistillBackward, GraphedDistillStep, ProjectorTrainer, distill_step


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="attention-distillation training of the alignment projector (HIP path)")
    p.add_argument("--pretrained_model_name_or_path", type=str, default=None, help="diffusers FLUX directory (transformer/ is read)")
    p.add_argument("--proj_path", type=str, default=None, help="projector state dict to start from (.bin)")
    p.add_argument("--mllm", type=str, default="3b", choices=["3b", "7b"], help="Qwen2.5-VL size: projector factory (:399-401)")
    p.add_argument("--output_dir", type=str, default="sdxl-model-finetuned")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--max_train_steps", type=int, default=200000)
    p.add_argument("--checkpointing_steps", type=int, default=500)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--lr_scheduler", type=str, default="constant", choices=["constant", "constant_with_warmup", "linear", "cosine"])
    p.add_argument("--lr_warmup_steps", type=int, default=500)
    p.add_argument("--adam_beta1", type=float, default=0.9)
    p.add_argument("--adam_beta2", type=float, default=0.999)
    p.add_argument("--adam_weight_decay", type=float, default=1e-2)
    p.add_argument("--adam_epsilon", type=float, default=1e-08)
    p.add_argument("--max_grad_norm", type=float, default=1.0)
    p.add_argument("--temperature", type=float, default=3.0, help="temperature0 of the distillation loss (:612)")
    p.add_argument("--local_infer_world_size", type=int, default=0, help="teacher ranks per node (0: every rank trains)")
    p.add_argument("--synthetic", action="store_true", help="random-init transformer and random batches at the reference's shapes")
    p.add_argument("--tiny", action="store_true", help="with --synthetic: reduced widths (tests)")
    p.add_argument("--batch_files", type=str, nargs="*", default=None, help="torch-saved batch dicts with the reference's keys")
    p.add_argument("--use_graph", action="store_true", help="replay one captured hipGraph per step (fixed batch shapes)")
    return p.parse_args(argv)


def lr_factor(name, step, warmup, total):
    """diffusers.optimization.get_scheduler factors for the schedules the reference's --lr_scheduler accepts here."""
    if name == "constant":
        return 1.0
    w = min(1.0, step / max(1, warmup)) if warmup > 0 else 1.0
    if name == "constant_with_warmup":
        return w
    if step < warmup:
        return w
    prog = (step - warmup) / max(1, total - warmup)
    if name == "linear":
        return max(0.0, 1.0 - prog)
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))  # cosine


def save_checkpoint(proj, output_dir, global_step):
    """<output_dir>/<global_step>/diffusion_pytorch_model.bin = the projector's state dict (train/train_qwenvl.py:643-648)."""
    path = os.path.join(output_dir, str(global_step))
    os.makedirs(path, exist_ok=True)
    torch.save({k: v.detach().cpu() for k, v in proj.state_dict().items()}, os.path.join(path, "diffusion_pytorch_model.bin"))
    return path


def latest_checkpoint(output_dir):
    """(step, path) of the highest-numbered <output_dir>/<step>/diffusion_pytorch_model.bin, or (0, None): the reference resumes from it
    and continues at global_step = that step (train/train_qwenvl.py:404-409, :535)."""
    best, path = 0, None
    if output_dir and os.path.isdir(output_dir):
        for d in os.listdir(output_dir):
            f = os.path.join(output_dir, d, "diffusion_pytorch_model.bin")
            if d.isdigit() and os.path.isfile(f) and int(d) > best:
                best, path = int(d), f
    return best, path


def synthetic_batch(bsz, device, gen, cfg, D, St, lat_hw, mllm_shape):
    """One batch with the reference's keys and (at full size) shapes, train/train_qwenvl.py:324-333."""
    Si = lat_hw * lat_hw
    C, H = mllm_shape
    rn = lambda *s: torch.randn(s, device=device, generator=gen)  # noqa: E731
    return dict(KD_teacher_tensor0=rn(bsz, cfg.num_layers, Si, D).bfloat16(), KD_teacher_tensor1=rn(bsz, cfg.num_layers, St, D).bfloat16(),
                KD_teacher_tensor2=rn(bsz, cfg.num_single_layers, St + Si, D).bfloat16(), latents=rn(bsz, Si, cfg.in_channels).bfloat16(),
                text_embeddings=(rn(bsz, C, St, H) * 3).bfloat16(), timestep=torch.randint(1, 1000, (bsz,), device=device, generator=gen).float())


def run(args):
    rank, world = xdist.init_from_env()
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    groups = None
    train_pg = None
    if world > 1 and args.local_infer_world_size > 0:
        groups = xdist.TeacherStudentGroups(rank, world, int(os.environ.get("LOCAL_WORLD_SIZE", world)), args.local_infer_world_size)
        train_pg = groups.train_pg
    gen = torch.Generator(device=device).manual_seed((args.seed or 0) + 1000 * rank)
    # ---- frozen student transformer (FLUX.1-dev layout: guidance embedding, :553-554) and the trainable projector
    if args.synthetic:
        if args.tiny:
            model = FluxTransformer2DModel(num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32,
                                           guidance_embeds=True, device=device).init_random_(seed=1)
            proj = Proj7Exp(in_channels=5, input_dim=128, output_dim0=32, output_dim1=64, use_t5=False, use_scale=False, use_cnn=True,
                            device=device).init_random_(2)
            St, lat_hw, mllm = 24, 8, (5, 128)
        else:
            model = FluxTransformer2DModel(guidance_embeds=True, device=device).init_random_(seed=1)
            fac = create_proj3_qwen3b if args.mllm == "3b" else create_proj3_qwen7b
            proj = fac(in_channels=37 if args.mllm == "3b" else 29, use_t5=False, use_scale=False, use_cnn=True, device=device).init_random_(2)
            St, lat_hw, mllm = 512, 64, ((37, 2048) if args.mllm == "3b" else (29, 3584))
    else:
        if not args.pretrained_model_name_or_path or not args.batch_files:
            raise SystemExit("train_distill: give --synthetic, or --pretrained_model_name_or_path and --batch_files")
        model = FluxTransformer2DModel.from_pretrained(args.pretrained_model_name_or_path, subfolder="transformer", device=device)
        if args.proj_path:
            from .checkpoints import load_projector_checkpoint
            proj = load_projector_checkpoint(args.proj_path, device=device, in_channels=37 if args.mllm == "3b" else 29)
        else:
            fac = create_proj3_qwen3b if args.mllm == "3b" else create_proj3_qwen7b
            proj = fac(in_channels=37 if args.mllm == "3b" else 29, use_t5=False, use_scale=False, use_cnn=True, device=device).init_random_(2)
        St, lat_hw, mllm = 512, 64, None
    D = model.inner_dim
    trainer = ProjectorTrainer(proj, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2), eps=args.adam_epsilon,
                               weight_decay=args.adam_weight_decay, max_grad_norm=args.max_grad_norm, process_group=train_pg)
    chain = DistillBackward(model)
    txt_ids = torch.zeros((St, 3), device=device)                                               # :551
    img_ids = FluxPipeline._prepare_latent_image_ids(1, lat_hw, lat_hw, device, torch.float32)   # :552
    guidance = torch.full((args.batch_size,), 3.5, device=device)                               # :553-554
    is_teacher = groups is not None and groups.is_infer_rank
    if groups is not None and not args.synthetic:
        # the teacher pipeline behind send_to_infer_device / receive_from_infer_device is not part of this program (teacher tensors come
        # pre-computed in --batch_files): refuse the layout instead of leaving GPUs idle without saying so
        raise SystemExit("--local_infer_world_size > 0 is only wired for --synthetic runs (no teacher pipeline in this program)")
    if is_teacher:
        # teacher ranks of a synthetic run have nothing to compute; they leave WITH the trainers (rank 0 is a teacher and, under a plain
        # env:// launch, hosts the store the trainers' first collective still needs)
        xdist.barrier_and_destroy()
        return []
    # resume (train/train_qwenvl.py:404-409, :447-459, :476-481, :535): the newest checkpoint under output_dir continues at its step
    # NUMBER, and that is all that continues -- the reference's checkpoint holds the projector weights only, so a restart builds a fresh
    # AdamW (moments zero, its own step count from 0: bias corrections and moments restart TOGETHER -- ADVICE r3: moments at zero under
    # a continued step count are uncorrected, 3-6x the nominal update for the first few hundred steps) and a fresh lr_scheduler (the
    # warm-up runs again).  trainer.step_count therefore stays 0 and the schedule below counts optimizer steps since the restart.
    global_step, losses, graphed = 0, [], None
    resume_step = 0
    last_step, last_path = latest_checkpoint(args.output_dir)
    if last_path is not None:
        from .checkpoints import load_projector_state_dict
        load_projector_state_dict(proj, last_path)
        global_step = resume_step = last_step
        if rank == 0 or (groups is not None and rank == groups.train_ranks[0]):
            print(f"resuming from {last_path} at global_step {global_step}", flush=True)
    step = global_step * args.gradient_accumulation_steps
    ga = args.gradient_accumulation_steps
    while global_step < args.max_train_steps:
        if args.synthetic:
            batch = synthetic_batch(args.batch_size, device, gen, model.config, D, St, lat_hw, mllm)
        else:
            batch = torch.load(args.batch_files[step % len(args.batch_files)], map_location=device)
        sync = step % args.gradient_accumulation_steps == 0                                      # :560
        # the reference builds its scheduler with warm-up and total steps multiplied by the accumulation count and steps it once per
        # optimizer step (train/train_qwenvl.py:476-481, :631): same factor sequence here
        trainer.lr = args.learning_rate * lr_factor(args.lr_scheduler, global_step - resume_step, args.lr_warmup_steps * ga, args.max_train_steps * ga)
        teacher = [batch["KD_teacher_tensor0"], batch["KD_teacher_tensor1"], batch["KD_teacher_tensor2"]]
        if args.use_graph:
            if graphed is None:
                graphed = GraphedDistillStep(trainer, chain, txt_ids, img_ids, guidance, args.temperature)
            loss = graphed(batch["text_embeddings"], batch["latents"], batch["timestep"] / 1000, teacher, optimizer_step=sync)
        else:
            loss = distill_step(trainer, chain, batch["text_embeddings"], batch["latents"], batch["timestep"] / 1000, teacher, txt_ids, img_ids,
                                guidance=guidance[: batch["latents"].shape[0]], temperature=args.temperature, optimizer_step=sync)
        step += 1
        if not args.use_graph:
            ops.streamk_poll()
        if sync:
            global_step += 1
            losses.append(float(loss))   # (synchronises: the marker reads enqueued above have completed)
            # a chained stream-K segment that gave up leaves undefined gradients: stop before another update or a checkpoint is built on them
            ops.streamk_check(sync=False)
            if rank == 0 or (groups is not None and rank == groups.train_ranks[0]):
                print(f"step {global_step}: step_loss {losses[-1]:.4f} lr {trainer.lr:.3e} grad_norm {float(trainer.last_norm[1]):.4e}", flush=True)
                if global_step % args.checkpointing_steps == 0:
                    ops.streamk_check(sync=True)
                    print("saving model to", save_checkpoint(proj, args.output_dir, global_step), flush=True)
    run.last = dict(trainer=trainer, resume_step=resume_step, global_step=global_step)   # (introspection for tests)
    if groups is not None:
        xdist.barrier_and_destroy()
    return losses


def main(argv=None):
    return run(parse_args(argv))


if __name__ == "__main__":
    main()
