"""textflux_amd -- the TextFlux / FLUX.1-Fill denoising path on MI355X (gfx950), hand-written HIP behind the
reference's own Python entry points.  See DESIGN.md / INTEGRATION.md."""


def __getattr__(name):  # lazy: importing the package must not import torch or load the .so
    if name in ("FluxFillPipeline", "FluxPipelineOutput"):
        from . import pipeline
        return getattr(pipeline, name)
    if name == "FluxTransformer2DModel":
        from .transformer import FluxTransformer2DModel
        return FluxTransformer2DModel
    if name in ("FlowMatchEulerDiscreteScheduler", "StochasticRFOvershotDiscreteScheduler"):
        from . import schedulers
        return getattr(schedulers, name)
    if name == "AutoencoderKL":
        from .vae import AutoencoderKL
        return AutoencoderKL
    raise AttributeError(name)
