"""x2i_amd -- MI355X-native implementation of the X2I sampling hot path.

Public surface mirrors the reference's Python call sites:
  x2i_amd.proj        create_proj3_qwen3b / create_proj3_qwen7b / create_proj_internvl1b / ... (utils/proj.py)
  x2i_amd.flux        FluxTransformer2DModel, ControlNeXtModel            (lightcontrol/lightcontrol_flux.py)
  x2i_amd.pipeline    FluxPipeline, FlowMatchEulerDiscreteScheduler       (diffusers call surface used by infer/*.py)
  x2i_amd.ops         one thin wrapper per C-ABI entry point (include/x2i.h)
"""
__version__ = "0.1.0"
