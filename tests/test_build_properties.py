"""Properties of the built gfx950 code that the measured speed depends on and that an innocent edit can lose without any test
noticing -- checked on the CPU by disassembling the code object inside the build (no GPU needed).

The extension kernel must have the whole walk inlined: with an out-of-line piece the Walk object's address escapes into the call
and the object moves to scratch memory, which tripled the kernel's memory instructions and cost 19 % of the default flow
("The Walk object out of scratch memory")."""
from __future__ import annotations

import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")
OBJ = ROOT / "longreadselfcorrect_amd" / "_build" / "obj" / "wp.hip.o"


def _disassemble(symbol: str, tmp: Path) -> list[str]:
    fat, co = tmp / "wp.fatbin", tmp / "wp.co"
    subprocess.run([str(LLVM / "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", str(OBJ)], check=True)
    subprocess.run([str(LLVM / "clang-offload-bundler"), "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--unbundle", f"--output={co}"], check=True)
    out = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", f"--disassemble-symbols={symbol}", str(co)],
                         check=True, capture_output=True, text=True).stdout
    return [l for l in out.splitlines() if re.match(r"^\s+[a-z_0-9]+\s", l)]


@pytest.mark.parametrize("wide", [False, True])
def test_extension_kernel_keeps_the_walk_object_out_of_memory(tmp_path, wide):
    import __graft_entry__ as g
    g.build()
    assert OBJ.exists(), "build() leaves the per-unit objects in _build/obj"
    sym = f"_ZN4lrsc16wp_extend_kernelILb{int(wide)}EEEvNS_10FmIndexDevENS_6WpArgsE"
    ins = _disassemble(sym, tmp_path)
    assert len(ins) > 20000, "the kernel with the whole walk inside is some 29 k instructions"
    calls = [l for l in ins if "s_swappc_b64" in l]
    assert not calls, "an out-of-line piece of the walk: the Walk object's address escapes into it and the object moves to scratch"
    n_scratch = sum("scratch_" in l for l in ins)
    n_flat = sum(re.match(r"^\s+flat_", l) is not None for l in ins)
    # measured at the time of writing: 364 scratch + 633 FLAT (narrow), against 1 012 + 1 864 with the pieces as calls
    assert n_scratch < 700 and n_flat < 1000, (n_scratch, n_flat)
