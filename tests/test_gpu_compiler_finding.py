"""The reduced form of the compiler finding (tools/repro_complement/repro.hip, profiles/r03_compiler_finding/README.md) on the GPU
box: the safe source forms (code formed with xor + mask; code masked before the select chain; base count picked by the code's two bits, which
is what rank_device.h does) must match the host
evaluation of the same source for every lane.  Whether the unguarded form (variant 0) still miscompiles is recorded, not asserted:
a fixed compiler is not a failure."""
from __future__ import annotations

import shutil
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
SRC = Path(__file__).resolve().parent.parent / "tools" / "repro_complement" / "repro.hip"


@pytest.mark.parametrize("variant", [1, 2, 3, 0])
def test_select_chain_over_block_counters(tmp_path, variant):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / f"repro{variant}"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", f"-DVARIANT={variant}", str(SRC), "-o", str(exe)], check=True,
                   capture_output=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    print(r.stdout.strip())
    if variant == 0:
        assert r.returncode in (0, 1), r.stdout + r.stderr          # 1 = the finding reproduces with this compiler; 0 = it no longer does
    else:
        assert r.returncode == 0, r.stdout + r.stderr
