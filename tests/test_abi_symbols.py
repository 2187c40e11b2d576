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
    """Nothing under ppo_amd/ imports, includes, links or loads anything of oracle/ (comments may
    name the oracle as the place where a CPU twin lives)."""
    import re
    bad = re.compile(r"(^\s*(import|from)\s+oracle\b|#\s*include\s*[\"<][^\">]*oracle|libppo_xp_oracle|oracle/|orc_[a-z_0-9]+\s*\()", re.M)
    pkg = os.path.join(ROOT, "ppo_amd")
    seen = 0
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                seen += 1
                m = bad.search(src)
                assert m is None, (f, m.group(0))
    assert seen > 8
