"""The reference's unit tests re-stated in C++ against include/ppo_xp.hpp (tests/cpp/): a compiled
host program calling libppo_xp.so the way a Rust extern "C" shim would."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_reference_cases():
    exe = os.path.join(ROOT, "tests", "cpp", "test_reference_cases")
    assert os.path.exists(exe), "run __graft_entry__.build() (make -C tests/cpp)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all reference cases passed" in r.stdout


def test_cpp_abi_stress():
    """tests/cpp/abi_stress.cpp: every entry point, several handles, error returns, from compiled code."""
    exe = os.path.join(ROOT, "tests", "cpp", "abi_stress")
    assert os.path.exists(exe), "run __graft_entry__.build() (make -C tests/cpp)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi stress passed" in r.stdout
