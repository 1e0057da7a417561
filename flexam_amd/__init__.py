"""flexam_amd -- MI355X-native implementation of FlexAM's denoising hot path.

Exports the reference's names (FlexAM/models/__init__.py, FlexAM/pipeline/__init__.py) so
`pipelines.py` / ComfyUI nodes can import them unchanged:
    Wan2_2Transformer3DModel_FlexAM, WanTransformer3DModel_FlexAM, AutoencoderKLWan3_8,
    Wan2_2FunControlPipeline_FlexAM, attention
and, for the step in front of the sampler, `visualize_tracking_DELTA` (pipelines.py:1852: tracks -> conditioning videos).
Arithmetic runs in libflexam_hip.so (hand-written gfx950 HIP kernels, C ABI in
include/flexam_hip.h); this package is the host-side mirror of the reference interface.
"""
__all__ = ["Wan2_2Transformer3DModel_FlexAM", "WanTransformer3DModel_FlexAM", "AutoencoderKLWan3_8",
           "Wan2_2FunControlPipeline_FlexAM", "FlowMatchEulerDiscreteScheduler", "FlowUniPCMultistepScheduler",
           "FlowDPMSolverMultistepScheduler", "WanT5EncoderModel", "attention", "visualize_tracking_DELTA"]


def __getattr__(name):
    # lazy: importing the package must not need torch.cuda or the built library
    if name in ("Wan2_2Transformer3DModel_FlexAM", "WanTransformer3DModel_FlexAM"):
        from . import wan_transformer3d_FlexAM as m
        return getattr(m, name)
    if name == "AutoencoderKLWan3_8":
        from .wan_vae3_8 import AutoencoderKLWan3_8
        return AutoencoderKLWan3_8
    if name in ("Wan2_2FunControlPipeline_FlexAM", "WanPipelineOutput"):
        from . import pipeline_wan2_2_fun_control_FlexAM as m
        return getattr(m, name)
    if name == "FlowMatchEulerDiscreteScheduler":
        from .scheduler import FlowMatchEulerDiscreteScheduler
        return FlowMatchEulerDiscreteScheduler
    if name == "FlowUniPCMultistepScheduler":
        from .fm_solvers_unipc import FlowUniPCMultistepScheduler
        return FlowUniPCMultistepScheduler
    if name == "FlowDPMSolverMultistepScheduler":
        from .fm_solvers import FlowDPMSolverMultistepScheduler
        return FlowDPMSolverMultistepScheduler
    if name == "WanT5EncoderModel":
        from .wan_text_encoder import WanT5EncoderModel
        return WanT5EncoderModel
    if name == "attention":
        from .attention_utils import attention
        return attention
    if name == "visualize_tracking_DELTA":
        from .conditioning_raster import visualize_tracking_DELTA
        return visualize_tracking_DELTA
    raise AttributeError(name)
