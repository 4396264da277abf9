"""Memory-safety fuzz of the code a GPU lane runs on untrusted bytes -- the DEFLATE lane decoder (csrc/ginflate.h) and
the BAM record parser (csrc/bamscan.h) -- as host builds under AddressSanitizer + UndefinedBehaviorSanitizer
(tests/native/*.cpp).  On the GPU an access outside a buffer is a memory fault that can take the machine down, and GPU
sanitizers are not available: the same sources are therefore driven with damaged and random input here, with every
buffer at its exact size and every per-lane table index checked against the sizes the device keeps."""

import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = shutil.which('clang++', path='/opt/rocm/lib/llvm/bin') or shutil.which('amdclang++')


def _build(name, tmp_path, libs=()):
    exe = str(tmp_path / name)
    cmd = [CLANG, '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
           '-I' + os.path.join(ROOT, 'te_counter_amd', 'csrc'), '-I' + os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'tests', 'native', name + '.cpp'), '-o', exe, *libs]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.skipif(CLANG is None, reason='needs the ROCm clang++ (the sources use its bit-reverse builtin)')
@pytest.mark.parametrize('name,iterations,libs', [('fuzz_ginflate', 8000, ('-lz',)), ('fuzz_bamscan', 60000, ())])
def test_sanitized_host_build_survives_damaged_input(name, iterations, libs, tmp_path):
    exe = _build(name, tmp_path, libs)
    for seed in (1, 2):
        r = subprocess.run([exe, str(iterations), str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert name in r.stdout
