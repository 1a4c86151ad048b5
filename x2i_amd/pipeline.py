"""Sampling pipeline with the call surface the reference's inference scripts use on diffusers' FluxPipeline
(infer/inference_qwenvl.py:72-73,188-217; restated semantics: SURVEY.md Appendix A.1):

    pipeline = FluxPipeline(transformer, scheduler)
    latents = pipeline(prompt_embeds=..., pooled_prompt_embeds=..., num_inference_steps=4, guidance_scale=3.5,
                       height=1024, width=1024, output_type="latent"[, generator=...][, latents=...]).images
    latents = FluxPipeline._unpack_latents(latents, height, width, vae_scale_factor)

The loop body (transformer + Euler update) runs entirely in libx2i_hip.so; with `use_graph=True` the whole N-step
loop of a shape that comes back is captured into a hipGraph and replayed (an LRU of graphs; a shape seen for the first time runs
eagerly and its pass is the answer) -- timesteps and sigma deltas are device-side inputs.
Batch > 1 and batch sharding over ranks (x2i_amd.dist) are this build's extension along the axis the API already has
(`prompt_embeds.shape[0]`).
"""
import math
from typing import Optional

import os

import numpy as np
import torch

from . import ops


class FlowMatchEulerDiscreteScheduler:
    """diffusers 0.31 FlowMatchEulerDiscreteScheduler (the subset FluxPipeline drives): set_timesteps(sigmas=, mu=)
    and step().  Config keys as in scheduler/scheduler_config.json of the checkpoint."""

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, use_dynamic_shifting: bool = False,
                 base_shift: float = 0.5, max_shift: float = 1.15, base_image_seq_len: int = 256,
                 max_image_seq_len: int = 4096):
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, shift=shift, use_dynamic_shifting=use_dynamic_shifting,
                           base_shift=base_shift, max_shift=max_shift, base_image_seq_len=base_image_seq_len,
                           max_image_seq_len=max_image_seq_len)
        ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = ts / num_train_timesteps
        if not use_dynamic_shifting:
            sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigma_min, self.sigma_max = float(sig[-1]), float(sig[0])
        self.timesteps = torch.from_numpy(sig * num_train_timesteps)
        self.sigmas = torch.from_numpy(sig)
        self._step_index = None

    @classmethod
    def from_config(cls, cfg):
        keys = ("num_train_timesteps", "shift", "use_dynamic_shifting", "base_shift", "max_shift", "base_image_seq_len",
                "max_image_seq_len")
        return cls(**{k: cfg[k] for k in keys if k in cfg})

    @staticmethod
    def time_shift(mu, sigma, t):
        return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None):
        c = self.config
        if c.use_dynamic_shifting and mu is None:
            raise ValueError(" you have a pass a value for `mu` when `use_dynamic_shifting` is set to be `True`")
        if sigmas is None:
            ts = np.linspace(self.sigma_max * c.num_train_timesteps, self.sigma_min * c.num_train_timesteps, num_inference_steps)
            sigmas = ts / c.num_train_timesteps
        sigmas = np.asarray(sigmas, dtype=np.float64)
        if c.use_dynamic_shifting:
            sigmas = self.time_shift(mu, 1.0, sigmas)
        else:
            sigmas = c.shift * sigmas / (1 + (c.shift - 1) * sigmas)
        sigmas = torch.from_numpy(np.asarray(sigmas)).to(dtype=torch.float32, device=device)
        self.timesteps = (sigmas * c.num_train_timesteps).to(device=device)
        self.sigmas = torch.cat([sigmas, torch.zeros(1, device=sigmas.device)])
        self._step_index = 0

    def step(self, model_output, timestep, sample, return_dict=False):
        """x <- bf16(f32(x) + (sigma_next - sigma) * eps), on the HIP path."""
        i = self._step_index
        dt = (self.sigmas[i + 1] - self.sigmas[i]).reshape(1).to(device=sample.device, dtype=torch.float32)
        if sample.dtype == torch.bfloat16 and model_output.dtype == torch.bfloat16:
            prev = ops.euler_step_(sample.clone(), model_output.contiguous(), dt)
        else:  # fp32 latents in, model dtype out (diffusers casts to model_output.dtype)
            prev = (sample.to(torch.float32) + dt * model_output).to(model_output.dtype)
        self._step_index += 1
        return (prev,)


def calculate_shift(image_seq_len, base_seq_len: int = 256, max_seq_len: int = 4096, base_shift: float = 0.5,
                    max_shift: float = 1.16):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


class FluxPipelineOutput:
    def __init__(self, images):
        self.images = images


class FluxPipeline:
    """`transformer`: x2i_amd.flux.FluxTransformer2DModel; `scheduler`: FlowMatchEulerDiscreteScheduler."""

    def __init__(self, transformer, scheduler=None, vae=None, control_nets=None):
        self.transformer = transformer
        self.scheduler = scheduler or FlowMatchEulerDiscreteScheduler()
        self.vae = vae
        self.vae_scale_factor = 16  # diffusers 0.31: 2 ** len(vae.config.block_out_channels) == 16 for FLUX
        self.default_sample_size = 64
        self.control_nets = control_nets
        # all steps' AdaLN tables in front of the loop (False / X2I_HOIST_MOD=0: one table per step, A/B; identical results)
        self.hoist_modulation = os.environ.get("X2I_HOIST_MOD", "1") != "0"
        # hipGraph cache: a key (batch, text length, size, steps, guidance, hint, fp8 mode) is run EAGERLY the first time it is seen and
        # that pass's latents are returned; it is captured when it comes back, and the captured graphs form an LRU bounded by count and
        # bytes (ragged prompt lengths -- infer/inference_minicpm.py:160-177, inference_multi_turn.py:132-156 -- neither pay two passes
        # per new length nor evict each other's graphs)
        self._graphs = {}           # key -> [graph, static inputs, aux tensors, bytes]; insertion order = recency
        self._seen = {}             # keys run eagerly once (a warm-up pass exists), not yet captured
        self.graph_cache_entries = int(os.environ.get("X2I_GRAPH_CACHE_ENTRIES", "8"))
        self.graph_cache_bytes = int(float(os.environ.get("X2I_GRAPH_CACHE_GB", "32")) * 2 ** 30)
        self.graph_stats = dict(eager=0, captures=0, replays=0, evictions=0)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, text_encoder=None, text_encoder_2=None, tokenizer=None,
                        tokenizer_2=None, vae=None, revision=None, torch_dtype=torch.bfloat16, transformer=None, scheduler=None,
                        **unused):
        """The reference's one construction call, unchanged (infer/inference_qwenvl.py:72-73 and the three sibling scripts):

            pipeline = FluxPipeline.from_pretrained(flux_path, text_encoder=None, text_encoder_2=None, tokenizer=None,
                                                    tokenizer_2=None, vae=None, revision="refs/pr/1", torch_dtype=dtype).to(device)

        Reads `<path>/transformer` (config.json + safetensors shards, diffusers key names) and
        `<path>/scheduler/scheduler_config.json` (never hard-coded: schnell / shuttle-3 / dev differ).  Text encoders and
        tokenizers are never used on this path (every script passes None); a `vae` object is kept as given.  `revision`
        is a hub concept and is ignored for a local directory.  Weights land on the current HIP device when one is visible
        (the HIP path has no CPU execution) and `.to(device)` moves them like the reference's call chain does."""
        import json
        import os
        for name, v in (("text_encoder", text_encoder), ("text_encoder_2", text_encoder_2), ("tokenizer", tokenizer),
                        ("tokenizer_2", tokenizer_2)):
            if v is not None:
                raise ValueError("x2i_amd FluxPipeline is driven by prompt_embeds: pass %s=None (as the reference does)" % name)
        if torch_dtype != torch.bfloat16:
            raise ValueError("x2i_amd: the HIP path computes in bf16 (torch_dtype=torch.bfloat16, the reference's dtype)")
        path = pretrained_model_name_or_path
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        if transformer is None:
            from .flux import FluxTransformer2DModel
            transformer = FluxTransformer2DModel.from_pretrained(path, subfolder="transformer", torch_dtype=torch_dtype, device=device)
        if scheduler is None:
            with open(os.path.join(path, "scheduler", "scheduler_config.json")) as fh:
                scheduler = FlowMatchEulerDiscreteScheduler.from_config(json.load(fh))
        return cls(transformer, scheduler, vae=vae)

    def to(self, device=None, dtype=None, **unused):
        if dtype not in (None, torch.bfloat16):
            raise ValueError("x2i_amd FluxPipeline is bf16-only")
        if device is not None:
            self.transformer.to(device)
            for n in (self.control_nets or []):
                n.to(device)
            if self.vae is not None and hasattr(self.vae, "to"):
                self.vae.to(device)
            self._graphs, self._seen = {}, {}
        return self

    @property
    def device(self):
        return self.transformer.device

    # ---- static helpers with the reference's names (copies: train/train_qwenvl.py:216-234, train_lightcontrol.py:403-410)
    @staticmethod
    def _pack_latents(latents, batch_size, num_channels_latents, height, width):
        latents = latents.view(batch_size, num_channels_latents, height // 2, 2, width // 2, 2)
        latents = latents.permute(0, 2, 4, 1, 3, 5)
        return latents.reshape(batch_size, (height // 2) * (width // 2), num_channels_latents * 4)

    @staticmethod
    def _unpack_latents(latents, height, width, vae_scale_factor):
        batch_size, num_patches, channels = latents.shape
        height = height // vae_scale_factor
        width = width // vae_scale_factor
        latents = latents.view(batch_size, height, width, channels // 4, 2, 2)
        latents = latents.permute(0, 3, 1, 4, 2, 5)
        return latents.reshape(batch_size, channels // (2 * 2), height * 2, width * 2)

    @staticmethod
    def _prepare_latent_image_ids(batch_size, height, width, device, dtype):
        """height/width are the PACKED grid sizes (diffusers 0.31 convention)."""
        ids = torch.zeros(height, width, 3)
        ids[..., 1] = ids[..., 1] + torch.arange(height)[:, None]
        ids[..., 2] = ids[..., 2] + torch.arange(width)[None, :]
        return ids.reshape(height * width, 3).to(device=device, dtype=dtype)

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        height = 2 * (int(height) // self.vae_scale_factor)
        width = 2 * (int(width) // self.vae_scale_factor)
        ids = self._prepare_latent_image_ids(batch_size, height // 2, width // 2, device, dtype)
        if latents is not None:
            return latents.to(device=device, dtype=dtype), ids
        shape = (batch_size, num_channels_latents, height, width)
        gdev = generator.device if generator is not None else device
        noise = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)  # randn_tensor semantics
        return self._pack_latents(noise, batch_size, num_channels_latents, height, width), ids

    @torch.no_grad()
    def __call__(self, prompt_embeds=None, pooled_prompt_embeds=None, num_inference_steps: int = 28,
                 guidance_scale: float = 3.5, height: Optional[int] = None, width: Optional[int] = None,
                 output_type: str = "latent", generator=None, latents=None, guided_hint=None, use_graph: bool = False,
                 return_dict: bool = True, **unused):
        if prompt_embeds is None or pooled_prompt_embeds is None:
            raise ValueError("x2i_amd FluxPipeline is driven by prompt_embeds/pooled_prompt_embeds (no text encoders), "
                             "as every reference inference script does")
        if output_type != "latent":
            raise ValueError('only output_type="latent" is on the hot path (VAE decode is the caller\'s, as in the reference)')
        tr = self.transformer
        ops.streamk_check(sync=False)  # a stream-K give-up marker read behind an earlier call fails THIS call loudly (never observed)
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        if height % 16 or width % 16:
            raise ValueError("height and width must be multiples of 16")
        device = tr.device
        dtype = prompt_embeds.dtype
        B = prompt_embeds.shape[0]
        prompt_embeds = prompt_embeds.to(device)
        pooled_prompt_embeds = pooled_prompt_embeds.to(device)
        txt_ids = torch.zeros(prompt_embeds.shape[1], 3, device=device, dtype=dtype)
        C = tr.config.in_channels // 4
        latents, img_ids = self.prepare_latents(B, C, height, width, dtype, device, generator, latents)
        sc = self.scheduler.config
        sigmas = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)
        mu = calculate_shift(latents.shape[1], sc.base_image_seq_len, sc.max_image_seq_len, sc.base_shift, sc.max_shift)
        self.scheduler.set_timesteps(sigmas=sigmas, device=device, mu=mu)
        timesteps = self.scheduler.timesteps
        guidance = None
        if tr.config.guidance_embeds:
            guidance = torch.full([1], guidance_scale, device=device, dtype=torch.float32).expand(B)
        hint = guided_hint.to(device) if (self.control_nets is not None and guided_hint is not None) else None
        latents = latents.contiguous()
        if latents.dtype != torch.bfloat16:
            # fp32 callers (parity tests): keep the scheduler arithmetic in fp32 like diffusers does
            state = tr.prepare_conditioning(prompt_embeds, pooled_prompt_embeds, txt_ids, img_ids, guidance)
            control = self._control_fn(hint)
            for i, t in enumerate(timesteps):
                ts = t.expand(B).to(latents.dtype)
                noise = tr.denoise(state, latents, ts / 1000, control=control)
                latents = self.scheduler.step(noise.to(latents.dtype), t, latents)[0]
            return FluxPipelineOutput(latents) if return_dict else (latents,)

        dts = (self.scheduler.sigmas[1:] - self.scheduler.sigmas[:-1]).to(device=device, dtype=torch.float32).contiguous()
        tvals = [(t.expand(B).to(torch.bfloat16) / 1000).contiguous() for t in timesteps]

        # every tensor the captured launches read lives in `aux` so that it stays alive (and in place) with the graph
        aux = dict(txt_ids=txt_ids, img_ids=img_ids, guidance=guidance, dts=dts, tvals=tvals,
                   ws=tr._workspace(B, prompt_embeds.shape[1], latents.shape[1]))

        def body(pe, pooled, lat, hnt):
            """prepare + N x (transformer, Euler) -- every launch goes to the current stream (graph-capturable)."""
            state = tr.prepare_conditioning(pe, pooled, aux["txt_ids"], aux["img_ids"], aux["guidance"])
            control = self._control_fn(hnt)
            # the schedule is known: several steps' AdaLN tables per pass over the modulation weights at batch 1 / 2 (prepare_modulation)
            mods = tr.prepare_modulation(state, tvals, lat.dtype) if self.hoist_modulation else [None] * len(tvals)
            for i in range(len(tvals)):
                noise = tr.denoise(state, lat, tvals[i], control=control, mod=mods[i])
                ops.euler_step_(lat, noise, dts[i:i + 1])

        key = (B, prompt_embeds.shape[1], height, width, num_inference_steps, float(guidance_scale), hint is not None,
               getattr(self.transformer, "_fp8_mode", None),  # a graph captured on the bf16 path must not serve the e4m3 path
               self.hoist_modulation,
               ops.option_epoch())   # x2i_set_option calls since import: options select kernels, and a kernel's first launch raises its dynamic-LDS
                                     # attribute (not capturable) -- an eager pass under the CURRENT options must come before a capture
        entry = self._graphs.get(key) if use_graph else None
        if entry is None and not (use_graph and key in self._seen):
            # eager launch sequence: the caller's choice, or a key seen for the FIRST time -- its pass IS the answer (and the warm-up of
            # every lazy allocation / kernel attribute of this shape); a graph is captured only when the key comes back
            latents = latents.clone()
            body(prompt_embeds, pooled_prompt_embeds, latents, hint)
            ops.streamk_poll()
            self.graph_stats["eager"] += 1
            self._seen[key] = True
            while len(self._seen) > 256:
                self._seen.pop(next(iter(self._seen)))
            return FluxPipelineOutput(latents) if return_dict else (latents,)
        if entry is None:
            # what the graph pins: static inputs + its workspace + the capture's private pool.  Measured as the growth of RESERVED memory around
            # all three (activations freed during the capture return to the graph's private pool: they lower memory_allocated but stay pinned),
            # with the cache emptied first so that the growth is this graph's and not a refill of cached blocks
            torch.cuda.synchronize(device)
            torch.cuda.empty_cache()
            before = torch.cuda.memory_reserved(device)
            static = dict(pe=prompt_embeds.clone(), pooled=pooled_prompt_embeds.clone(), lat=latents.clone(),
                          hint=None if hint is None else hint.clone())
            # the graph owns its stream-K workspace (ops.streamk_scope): two graphs replayed on two streams never share one
            aux["sk_ws"] = ops.StreamKWorkspace(device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), ops.streamk_scope(aux["sk_ws"]):
                body(static["pe"], static["pooled"], static["lat"], static["hint"])
            torch.cuda.synchronize(device)
            nbytes = max(0, torch.cuda.memory_reserved(device) - before)
            entry = [graph, static, aux, nbytes]  # ids / guidance / schedule tensors must outlive the call: the graph reads them
            self._graphs[key] = entry
            self.graph_stats["captures"] += 1
            self._evict(keep=key)
        else:
            self._graphs[key] = self._graphs.pop(key)   # most recently used last
        graph, static = entry[0], entry[1]
        static["pe"].copy_(prompt_embeds)
        static["pooled"].copy_(pooled_prompt_embeds)
        static["lat"].copy_(latents)
        if hint is not None:
            static["hint"].copy_(hint)
        graph.replay()
        entry[2]["sk_ws"].poll()
        self.graph_stats["replays"] += 1
        out = static["lat"].clone()
        return FluxPipelineOutput(out) if return_dict else (out,)

    def _evict(self, keep=None):
        """LRU over the captured graphs, bounded by entries AND by the device bytes they pin (static inputs, stream-K workspace, the
        activations in the capture's private pool: ~2.5 GB per 1024^2 batch-4 graph)."""
        def total():
            return sum(e[3] for e in self._graphs.values())
        while len(self._graphs) > 1 and (len(self._graphs) > self.graph_cache_entries or total() > self.graph_cache_bytes):
            victim = next(k for k in self._graphs if k != keep)
            del self._graphs[victim]
            self.graph_stats["evictions"] += 1

    def _control_fn(self, hint):
        if hint is None:
            return None
        from .lightcontrol import make_control_fn
        return make_control_fn(self.control_nets, hint)


class VaeImageProcessor:
    """diffusers.image_processor.VaeImageProcessor(vae_scale_factor).postprocess(image, output_type=) as the reference uses it
    after vae.decode (infer/inference_qwenvl.py:210,216): denormalise [-1,1] -> [0,1], clamp, NCHW -> NHWC uint8 -> PIL."""

    def __init__(self, vae_scale_factor: int = 8, do_normalize: bool = True):
        self.vae_scale_factor = vae_scale_factor
        self.do_normalize = do_normalize

    @staticmethod
    def denormalize(images):
        return (images / 2 + 0.5).clamp(0, 1)

    def postprocess(self, image, output_type: str = "pil"):
        image = image.detach().float()
        if self.do_normalize:
            image = self.denormalize(image)
        if output_type == "pt":
            return image
        arr = image.cpu().permute(0, 2, 3, 1).numpy()
        if output_type == "np":
            return arr
        if output_type != "pil":
            raise ValueError("output_type must be 'pil', 'np' or 'pt'")
        from PIL import Image
        return [Image.fromarray(a) for a in (arr * 255).round().astype("uint8")]


class _Cfg(dict):
    __getattr__ = dict.__getitem__
