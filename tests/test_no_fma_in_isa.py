"""The bit-exactness of the GAE path rests on one compiler flag (-ffp-contract=off, ppo_amd/csrc/Makefile): every fp32
operation of buffer.rs:217-222 / data.rs:62 must be rounded on its own, as ndarray does.  This test disassembles the
gfx950 code objects inside the freshly built libppo_xp.so (no GPU needed) and asserts that the kernels which carry that
arithmetic contain no fused fp32 multiply-add of any flavour."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# kernels whose fp32 arithmetic must match the reference bit for bit
EXACT = ("gae_scan_kernel", "gae_wide_kernel", "gae_onepass_kernel", "unpack_flat_kernel", "pack_records_kernel", "fused_gather_kernel",
         "scalar_gather_kernel", "moments_kernel", "rnorm_accumulate_kernel")
FUSED = re.compile(r"\bv_(fma|fmac|mad|mac|pk_fma|fmaak|fmamk|madak|madmk)_(f32|legacy_f32|f16|f64)(_|\b)|\bv_dot", re.I)


def _disassemble(tmp_path):
    lib = os.path.join(ROOT, "ppo_amd", "libppo_xp.so")
    if not (os.path.exists(lib) and os.path.exists(OBJDUMP)):
        pytest.skip("library or llvm-objdump missing")
    work = tmp_path / "isa"
    work.mkdir()
    shutil.copy(lib, work / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
    text = ""
    for f in sorted(os.listdir(work)):
        if "gfx950" in f:
            text += subprocess.run([OBJDUMP, "-d", f], cwd=work, check=True, capture_output=True, text=True).stdout
    return text


def _functions(text):
    out, name = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m:
            name = m.group(1)
            out.setdefault(name, [])
        elif name is not None:
            out[name].append(ln)
    return out


def test_exact_kernels_have_no_fused_multiply_add(tmp_path):
    fns = _functions(_disassemble(tmp_path))
    assert len(fns) > 20, "no gfx950 code found inside libppo_xp.so"
    seen = set()
    for name, body in fns.items():
        tag = next((k for k in EXACT if k in name), None)
        if tag is None:
            continue
        seen.add(tag)
        bad = [ln.strip() for ln in body if FUSED.search(ln)]
        assert not bad, f"{name}: fused multiply-add in a kernel that must round every op: {bad[:3]}"
        assert any(re.search(r"\bv_(mul|add|sub)_f32", ln) for ln in body) or "pack_records" in name, name
    must = {"gae_scan_kernel", "gae_wide_kernel", "unpack_flat_kernel", "fused_gather_kernel", "scalar_gather_kernel", "rnorm_accumulate_kernel"}
    assert must <= seen, must - seen


def test_the_check_can_fail():
    assert FUSED.search("  v_fma_f32 v1, v2, v3, v4") and FUSED.search("v_fmac_f32_e32 v0, v1, v2") and FUSED.search("v_mad_f32 v0, v1, v2, v3")
    assert not FUSED.search("v_mul_f32_e32 v0, v1, v2") and not FUSED.search("v_mad_u64_u32 v[0:1], s[0:1], v2, v3, v[4:5]")
