"""The C-ABI shared library loads without a GPU and exports every symbol the header declares."""
import ctypes
import os
import re

from agplace_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "agplace_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(agp_[a-z0-9_]+)\s*\(", txt)))


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported_and_bound():
    names = header_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/agplace_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_identification_and_pure_host_entry_points():
    L = _lib.load()
    assert L.agp_arch() == b"gfx950"
    assert L.agp_version().startswith(b"agplace_hip")
    assert L.agp_knn_pad_rows(100000) % 128 == 0 and L.agp_knn_pad_rows(100000) >= 100000
    assert L.agp_knn_pad_rows(1) == 128
    assert L.agp_pool_workspace_floats(4, 256, 14, 84) > 0
    assert L.agp_knn_workspace_bytes(4096, 100000, 256, 20) > 4096 * 6250 * 4
    # a tiny database: the fp16 coarse pass writes TWO planes of [nq][32] words (its 64-row groups round up to 32 like the 16-row
    # ones), which must fit behind the query planes and the generic pass's minima (round 5: they did not, 77 KB out of bounds)
    assert L.agp_knn_workspace_bytes(600, 384, 256, 64) >= 2 * 600 * 256 * 2 + 24 * 608 * 4 + 2 * 600 * 32 * 4


def test_conv_desc_layout_matches_header():
    # 10 pointers + 16 int32, then the optional w_q8 pointer + its int32 exponent (+4 bytes of padding), stat_partial, and the
    # conv-epilogue pooling request (2 pointers + float + int32), and the optional chunk-major weight plane
    # ... the weight gradient's fp16 operand plane + gradient maxima, and the pooling statistic selector (+ 4 reserved bytes)
    assert ctypes.sizeof(_lib.ConvDesc) == 10 * 8 + 16 * 4 + 8 + 8 + 8 + 8 + 8 + 4 + 4 + 8 + 5 * 8 + 8 + 2 * 8 + 2 * 4
    assert _lib.ConvDesc.hi_only.offset == 188 and _lib.ConvDesc.pool_stat.offset == 264
    assert _lib.ConvDesc.w_cm.offset == 192 and _lib.ConvDesc.bstat_z_hi.offset == 200 and _lib.ConvDesc.bstat_rstd.offset == 232 and _lib.ConvDesc.w_cm_lo.offset == 240
    assert _lib.ConvDesc.in_h16.offset == 248 and _lib.ConvDesc.out_absmax.offset == 256
    assert _lib.ConvDesc.w_q8.offset == 144 and _lib.ConvDesc.w_q8_exp.offset == 152 and _lib.ConvDesc.stat_partial.offset == 160
    assert _lib.ConvDesc.pool_partial.offset == 168 and _lib.ConvDesc.pool_p.offset == 176 and _lib.ConvDesc.pool_eps.offset == 184
    # the same layout as the C compiler's, read from a tiny program compiled against the header
    import os, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "l.c")
        open(src, "w").write('#include <stddef.h>\n#include <stdio.h>\n#include "agplace_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                             'sizeof(agp_conv_desc),offsetof(agp_conv_desc,stat_partial),offsetof(agp_conv_desc,pool_partial),'
                             'offsetof(agp_conv_desc,pool_eps),offsetof(agp_conv_desc,w_cm),offsetof(agp_conv_desc,bstat_rstd),offsetof(agp_conv_desc,out_absmax),offsetof(agp_conv_desc,hi_only),offsetof(agp_conv_desc,pool_stat));return 0;}\n')
        exe = os.path.join(td, "l")
        subprocess.run(["gcc", "-I", os.path.join(root, "include"), src, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [ctypes.sizeof(_lib.ConvDesc), 160, 168, 184, 192, 232, 256, 188, 264]


def test_gp_buffer_size_agrees_across_header_kernels_and_host():
    """AGP_GP_FLOATS (the dL/dp buffer of the GeM backward entries: result + per-block partials + ticket) is spelled three times:
    include/agplace_hip.h, csrc/common.hpp (AGP_GP_SLOTS) and agplace_amd/_lib.py (GP_FLOATS)."""
    import re
    from agplace_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "agplace_hip.h")).read()
    m = re.search(r"#define AGP_GP_FLOATS \(1 \+ (\d+) \+ 1\)", hdr)
    assert m, "AGP_GP_FLOATS not found in the header"
    slots = int(m.group(1))
    common = open(os.path.join(root, "agplace_amd", "csrc", "common.hpp")).read()
    m2 = re.search(r"constexpr int AGP_GP_SLOTS = (\d+);", common)
    assert m2 and int(m2.group(1)) == slots
    assert _lib.GP_FLOATS == 1 + slots + 1
