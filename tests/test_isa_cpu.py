"""A mechanical fence around the one compiler behaviour that made a stage kernel run-to-run irreproducible ("the SLP hazard").

hipcc (ROCm 7.2) SLP-packs the combine of the per-wave LayerNorm partials of two token tiles into v_pk_add_f32 / v_pk_fma_f32 when it is left alone, and that build of
`sstage_kernel<4>` differed between two launches of the same inputs when two workgroups shared a CU (csrc/sstage.hip: SS_LN_OPAQUE2).  The source keeps the combine scalar with
opaque copies; this test makes sure it STAYS scalar: it cross-compiles the two stage kernels to gfx950 assembly here (no GPU needed, ~20 s) and fails on packed fp32
multiplies / FMAs or SGPR-pair splats between the `; LN_FENCE_BEGIN` / `; LN_FENCE_END` markers that bracket the statistics and their combine in every inlined copy of the LayerNorm (the normalisation behind them packs two channels of ONE token, which every bit-stable build has done)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lemevit_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
PACKED = re.compile(r"^\s*v_pk_(add|fma|mul)_f32\b")
# What may appear between the markers: v_pk_add_f32 on VGPR pairs only -- LLVM sinks the bias additions of the preceding projection (two CHANNELS of one token in adjacent
# registers) to their first use, the statistics loop; every bit-stable build has those.  What the irreproducible build had and must not come back: packed multiplies / FMAs
# (mean = sum * 1/NW, dev += d * d, M2 * 1/C + eps of TWO TOKEN TILES per instruction) and packed operands splatted from SGPR pairs.
def _hazard(line: str) -> bool:
    t = line.strip()
    return t.startswith(("v_pk_fma_f32", "v_pk_mul_f32")) or bool(re.search(r"\bs\[\d+:\d+\]", t.split(";")[0]))


def _asm(src: str, tmp_path) -> str:
    out = str(tmp_path / (src + ".s"))
    # the flags of csrc/Makefile for these two files
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fno-honor-nans", "-fno-slp-vectorize", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with open(out) as f:
        return f.read()


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc in this environment")
@pytest.mark.parametrize("src,min_regions", [("sstage.hip", 4), ("dstage.hip", 20)])
def test_layernorm_of_the_stage_kernels_stays_scalar(src, min_regions, tmp_path):
    text = _asm(src, tmp_path)
    inside, regions, bad, packed_elsewhere = False, 0, [], 0
    for no, line in enumerate(text.split("\n"), 1):
        if "; LN_FENCE_BEGIN" in line:
            assert not inside, f"{src}:{no}: nested LN_FENCE_BEGIN (a marker was moved by the compiler?)"
            inside, regions = True, regions + 1
        elif "; LN_FENCE_END" in line:
            assert inside, f"{src}:{no}: LN_FENCE_END without BEGIN"
            inside = False
        elif PACKED.match(line):
            if inside and _hazard(line):
                bad.append((no, line.strip()))
            elif not inside:
                packed_elsewhere += 1
    assert not inside
    assert regions >= min_regions, f"{src}: only {regions} fenced LayerNorm regions found in the assembly (expected >= {min_regions}): the markers are gone"
    assert packed_elsewhere > 0, f"{src}: no packed fp32 instruction anywhere (the GELU polynomial uses them on purpose): the pattern of this test no longer matches the ISA"
    assert not bad, f"{src}: packed fp32 arithmetic inside a LayerNorm of a stage kernel (run-to-run irreproducible on gfx950, csrc/sstage.hip): {bad[:5]} ... {len(bad)} in all"
