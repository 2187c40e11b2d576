"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/ppo_xp.h declares; the product path fails loudly (no CPU fallback) without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ppo_xp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ppo_xp_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from ppo_amd import _lib
    assert _declared() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from ppo_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(L, name), name
    assert b"gfx950" in _lib.load().ppo_xp_version()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import ppo_amd
    with pytest.raises(ppo_amd.PpoXpError) as e:
        ppo_amd.ExperienceBuffer(2, 3, 3)
    assert e.value.code == -19


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "ppo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower() or f == "synth.py", f
