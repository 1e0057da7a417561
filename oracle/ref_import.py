"""TEST INFRASTRUCTURE ONLY -- loads the *reference* FlexAM modules for fixture generation.

Runs only in the build container, where ``/root/reference`` is mounted.  Nothing
from the reference travels to the GPU box: this module is used by
``oracle/make_golden.py`` (to write ``tests/golden/*.safetensors``) and by the
``needs_reference`` CPU tests (live oracle-vs-reference comparison).  It is never
imported by ``flexam_amd`` (the product), ``bench.py`` or ``-m gpu`` tests.

Why stubs are needed:
  * ``diffusers`` is not installed and there is no network; the DiT / VAE classes
    only use 8 trivial symbols from it (mixins, a config decorator, 3 output
    holders).
  * ``FlexAM/dist`` is absent from the reference (swallowed by its .gitignore);
    the DiT imports 5 names from it at module import time
    (FlexAM/models/wan_transformer3d_FlexAM.py:22-24).
The reference ``.py`` files are loaded *by path* so the heavy package
``__init__`` files (which import missing third-party packages) never run.
"""
import functools
import importlib.util
import inspect
import logging
import os
import sys
import types

import torch.nn as nn

REF_ROOT = os.environ.get("FLEXAM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "FlexAM/models/wan_transformer3d_FlexAM.py"))


class _Config(dict):
    """`.config.patch_size` and `.config.get("add_ref_conv")` both work."""
    __getattr__ = dict.get


def _register_to_config(init):
    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        bound = inspect.signature(init).bind(self, *args, **kwargs)
        bound.apply_defaults()
        self.config = _Config({k: v for k, v in bound.arguments.items() if k != "self"})
        init(self, *args, **kwargs)
    return wrapped


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    class ConfigMixin:
        pass

    class ModelMixin(nn.Module):
        pass

    class FromOriginalModelMixin:
        pass

    class DecoderOutput:
        def __init__(self, sample):
            self.sample = sample

    class DiagonalGaussianDistribution:
        def __init__(self, parameters):
            self.mean, self.logvar = parameters.chunk(2, dim=1)

        def mode(self):
            return self.mean

    class AutoencoderKLOutput:
        def __init__(self, latent_dist):
            self.latent_dist = latent_dist

        def __getitem__(self, i):
            return (self.latent_dist,)[i]

    class _Logging:
        get_logger = staticmethod(logging.getLogger)

    for pkg in ("diffusers", "diffusers.loaders", "diffusers.models", "diffusers.models.autoencoders"):
        _module(pkg)
    _module("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=_register_to_config)
    _module("diffusers.loaders.single_file_model", FromOriginalModelMixin=FromOriginalModelMixin)
    _module("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _module("diffusers.utils", is_torch_version=lambda *a: True, logging=_Logging(), deprecate=lambda *a, **k: None,
            is_scipy_available=lambda: False)
    _module("diffusers.schedulers")

    class SchedulerMixin:
        pass

    class SchedulerOutput:
        def __init__(self, prev_sample):
            self.prev_sample = prev_sample

    _module("diffusers.schedulers.scheduling_utils", KarrasDiffusionSchedulers=[], SchedulerMixin=SchedulerMixin,
            SchedulerOutput=SchedulerOutput)

    def _randn_tensor(shape, generator=None, device=None, dtype=None):
        import torch
        return torch.randn(shape, generator=generator, device=device, dtype=dtype)

    _module("diffusers.utils.torch_utils", randn_tensor=_randn_tensor)
    _module("diffusers.models.autoencoders.vae", DecoderOutput=DecoderOutput,
            DiagonalGaussianDistribution=DiagonalGaussianDistribution)
    _module("diffusers.models.modeling_outputs", AutoencoderKLOutput=AutoencoderKLOutput)
    _module("diffusers.utils.accelerate_utils", apply_forward_hook=lambda f: f)

    for pkg in ("FlexAM", "FlexAM.models", "FlexAM.utils"):
        _module(pkg).__path__ = [os.path.join(REF_ROOT, pkg.replace(".", "/"))]

    def _absent(*a, **k):
        raise RuntimeError("FlexAM.dist is absent from the reference checkout")

    _module("FlexAM.dist", get_sequence_parallel_rank=_absent, get_sequence_parallel_world_size=_absent,
            get_sp_group=_absent, usp_attn_forward=_absent, xFuserLongContextAttention=_absent)


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_CACHE = {}


def load_reference():
    """Returns a namespace with the reference modules: .dit .vae .att .cache .cfg .unipc .dpm .t5"""
    if "ns" in _CACHE:
        return _CACHE["ns"]
    if not reference_available():
        raise RuntimeError(f"reference checkout not found under {REF_ROOT}")
    _install_stubs()
    cfg = _load("FlexAM.utils.cfg_optimization", "FlexAM/utils/cfg_optimization.py")
    sys.modules["FlexAM.utils"].cfg_skip = cfg.cfg_skip
    att = _load("FlexAM.models.attention_utils", "FlexAM/models/attention_utils.py")
    cache = _load("FlexAM.models.cache_utils", "FlexAM/models/cache_utils.py")
    _load("FlexAM.models.wan_camera_adapter", "FlexAM/models/wan_camera_adapter.py")
    dit = _load("FlexAM.models.wan_transformer3d_FlexAM", "FlexAM/models/wan_transformer3d_FlexAM.py")
    vae = _load("FlexAM.models.wan_vae3_8", "FlexAM/models/wan_vae3_8.py")
    t5 = _load("FlexAM.models.wan_text_encoder", "FlexAM/models/wan_text_encoder.py")
    unipc = _load("FlexAM.utils.fm_solvers_unipc", "FlexAM/utils/fm_solvers_unipc.py")
    dpm = _load("FlexAM.utils.fm_solvers", "FlexAM/utils/fm_solvers.py")
    ns = types.SimpleNamespace(dit=dit, vae=vae, att=att, cache=cache, cfg=cfg, unipc=unipc, dpm=dpm, t5=t5)
    _CACHE["ns"] = ns
    return ns
