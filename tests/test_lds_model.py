"""The LDS bank model behind the staging-image strides of the stage kernels (tools/lds_conflicts.py restates the LDS table of MI355X_MICROARCH.md).  CPU only: the
strides the kernels are built with must stay conflict-free under that model, and the packed strides of rounds 1 - 4 must show the 4-way conflict that was measured (the regression this
guards: somebody "tidying" SS_STG_ROW back to 96 or DS_STG_ENTRY back to 32)."""
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from lds_conflicts import cycles          # noqa: E402

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lemevit_amd", "csrc")


def _define(path, name):
    m = re.search(r"#define\s+%s\s+(\d+)" % name, open(os.path.join(CSRC, path)).read())
    assert m, f"{name} not found in {path}"
    return int(m.group(1))


def _tap_read(stride):          # one access of a ds_read2_b64: lane (g, li) reads 8 bytes of entry li at byte 8 g
    return cycles("r2x64", lambda l: (l & 15) * stride + (l >> 4) * 8)


def test_staging_strides_of_the_sources_are_conflict_free():
    for path, name, packed in (("sstage.hip", "SS_STG_ROW", 96), ("dstage.hip", "DS_STG_ENTRY", 32)):
        stride = _define(path, name)
        assert stride % 8 == 0 and stride >= packed
        got, groups = _tap_read(stride)
        assert got == groups == 4, (path, stride, got)
        assert _tap_read(packed)[0] == 16          # the packed entry: every 16-lane group on 4 banks


def test_model_reproduces_the_tabulated_cases():
    # lane-linear 16-byte reads (fragment order) and 8-byte reads are conflict-free; 64-byte rows read 16 bytes per lane by 16 consecutive rows are not
    assert cycles("r128", lambda l: l * 16) == (4, 4)
    assert cycles("r64", lambda l: l * 8) == (2, 2)
    assert cycles("r128", lambda l: (l & 15) * 64 + (l >> 4) * 16)[0] > 4
    # the attention backward's transposer: 32-byte key rows, 8-byte pieces; XOR-swizzled by row >> 2 the stores are conflict-free and the transposing read stays so
    assert cycles("w64", lambda l: (l & 15) * 32 + (l >> 4) * 8)[0] == 16
    assert cycles("w64", lambda l: (l & 15) * 32 + (((l >> 4) ^ ((l >> 2) & 3)) * 8))[0] == 4
    assert cycles("tr64", lambda l: ((l >> 4) * 4 + ((l & 15) >> 2)) * 32 + (((l & 3) ^ (l >> 4)) * 8)) == (2, 2)
