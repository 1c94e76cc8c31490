"""Oracle (test infrastructure, BUILD CONTAINER ONLY): import the reference's own
``src/network/CleanUMamba.py`` verbatim from /root/reference.

The reference cannot be imported as is: ``mamba_ssm`` and
``torchinfo`` are absent and ``src/util/util.py:220-227`` calls ``.cuda()`` in a
default argument at import time.  This module registers ``sys.modules`` stand-ins
for exactly those absent third-party names -- the Mamba block is supplied by
``oracle/mamba_ref.py`` -- and then imports the reference class unmodified, so
encoder / decoder / GLU / padding / normalisation / skip / streaming bookkeeping
executed through it are the reference's own code.  Nothing is copied.

Used only by ``oracle/make_golden.py``.  /root/reference does not exist on the GPU
box; nothing at test/bench time imports this file unless REFERENCE_ROOT exists.
"""
import os
import sys
import types

import torch

from . import mamba_ref

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "network"))


def load_reference():
    """Return the reference's ``src.network.CleanUMamba`` module."""
    if not available():
        raise RuntimeError("reference checkout not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # src/util/util.py:220-227 builds a loss module with .cuda() at import time.
    torch.nn.Module.cuda = lambda self, *a, **k: self

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "mamba_ssm" not in sys.modules:
        _mod("mamba_ssm")
        _mod("mamba_ssm.models")
        _mod("mamba_ssm.models.mixer_seq_simple", create_block=mamba_ref.create_block,
             _init_weights=mamba_ref._init_weights)
        _mod("mamba_ssm.utils")
        _mod("mamba_ssm.utils.generation", InferenceParams=mamba_ref.InferenceParams)
    if "torchinfo" not in sys.modules:
        _mod("torchinfo", summary=lambda *a, **k: None)
    import importlib
    return importlib.import_module("src.network.CleanUMamba")
