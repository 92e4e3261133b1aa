"""CPU-side check of the built gfx950 code objects inside libddn.so: the register-resident kernels (one wavefront
holds a whole 4096-point sequence in registers) are written so that nothing of that sequence is spilled.  The
compiler's register allocation is steered by source-level devices (opaque offsets, pins, a raised unroll threshold),
which a toolchain update could silently undo; this test reads the kernels' metadata notes and fails when a hot
kernel's private segment (scratch memory per lane) grows beyond what is recorded here."""
import os
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "convex_optimization_amd", "libddn.so")

msgpack = pytest.importorskip("msgpack")


def _kernels_of(path):
    """name -> metadata dict of every AMDGPU kernel in the ELF code objects embedded in `path`."""
    data = open(path, "rb").read()
    out = {}
    pos = 0
    while True:
        off = data.find(b"\x7fELF", pos)
        if off < 0:
            break
        pos = off + 4
        if data[off + 4] != 2 or len(data) < off + 64:
            continue
        e_machine = struct.unpack_from("<H", data, off + 18)[0]
        if e_machine != 224:        # EM_AMDGPU
            continue
        e_shoff = struct.unpack_from("<Q", data, off + 0x28)[0]
        e_shentsize, e_shnum = struct.unpack_from("<HH", data, off + 0x3A)
        for i in range(e_shnum):
            sh = off + e_shoff + i * e_shentsize
            if sh + 64 > len(data):
                break
            sh_type = struct.unpack_from("<I", data, sh + 4)[0]
            if sh_type != 7:        # SHT_NOTE
                continue
            sh_offset, sh_size = struct.unpack_from("<QQ", data, sh + 0x18)
            p, end = off + sh_offset, off + sh_offset + sh_size
            while p + 12 <= end:
                namesz, descsz, ntype = struct.unpack_from("<III", data, p)
                p += 12
                name = data[p:p + namesz].rstrip(b"\0")
                p += (namesz + 3) & ~3
                desc = data[p:p + descsz]
                p += (descsz + 3) & ~3
                if name == b"AMDGPU" and ntype == 32:       # NT_AMDGPU_METADATA (msgpack)
                    md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                    for k in md.get("amdhsa.kernels", []):
                        out[k[".name"]] = k
    return out


@pytest.fixture(scope="module")
def kernels():
    assert os.path.exists(LIB), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    ks = _kernels_of(LIB)
    assert len(ks) > 50, f"found only {len(ks)} kernels in the embedded code objects"
    return ks


def _family(kernels, token):
    fam = {n: k for n, k in kernels.items() if token in n}
    assert fam, f"no kernel named *{token}* in libddn.so"
    return fam


@pytest.mark.parametrize("token", ["k_wave_rowI", "k_wave_cpI", "k_wave_rowNI", "k_wave_cpNI"])
def test_register_resident_kernels_do_not_spill(kernels, token):
    """Row kernels (W = 4096 and 512-2048) and the Chambolle-Pock row kernels: zero scratch."""
    bad = {n: (k[".private_segment_fixed_size"], k.get(".vgpr_spill_count")) for n, k in _family(kernels, token).items()
           if k[".private_segment_fixed_size"] != 0}
    assert not bad, bad


@pytest.mark.parametrize("token", ["k_wave_colILi", "k_wave_colNILi"])
def test_wave_column_kernel_scratch_is_bounded(kernels, token):
    """The column kernels (H = 4096; H = 512-2048): the multiplier-mode variants use no scratch; the resolvent-mode
    variants park at most two dozen registers around the pointwise stage (stored once, reloaded once per workgroup: not
    in any inner loop).  A larger private segment means the transform's registers are being spilled."""
    for n, k in _family(kernels, token).items():
        res2 = (token + "0E") in n              # MODE = COL_RES2
        limit = 96 if res2 else 0
        assert k[".private_segment_fixed_size"] <= limit, (n, k[".private_segment_fixed_size"], k.get(".vgpr_spill_count"))
        assert k[".vgpr_count"] <= 256, (n, k[".vgpr_count"])


def test_two_waves_per_simd(kernels):
    """Every register-resident kernel must fit two wavefronts per SIMD (256 registers each): one more register and the
    CU holds four rows or columns instead of eight."""
    for token in ["k_wave_rowI", "k_wave_cpI", "k_wave_rowNI", "k_wave_cpNI", "k_wave_colILi", "k_wave_colNI"]:
        for n, k in _family(kernels, token).items():
            assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 256, (n, k[".vgpr_count"], k.get(".agpr_count"))


def test_whole_row_kernels_fit_their_occupancy(kernels):
    """The whole-row kernels (rows of 8192 / 16384 / 32768 pixels, csrc/ddn_wholerow.hpp) are compiled for two
    wavefronts per SIMD: at most 256 registers, or a 512-thread workgroup does not fit a compute unit and the
    256-thread ones run one row per CU instead of two.  The policy kernels park loop-invariant addresses and part of
    the row around the transforms (scratch outside the transform passes; bounded here), the Chambolle-Pock ones none."""
    fam = _family(kernels, "k_row_whole")
    assert len(fam) >= 12, sorted(fam)
    for n, k in fam.items():
        assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 256, (n, k[".vgpr_count"])
        limit = 0 if "k_row_whole_cp" in n else 768
        assert k[".private_segment_fixed_size"] <= limit, (n, k[".private_segment_fixed_size"], k.get(".vgpr_spill_count"))


def test_lds_row_kernels_scratch_budget(kernels):
    """The workgroup-per-row LDS kernels (csrc/ddn_kernels.hpp).  Those with the compile-time transform (the eighth
    template argument of k_solver_row, the fifth of k_cp_row: the row length) fit their registers: at most 32 B of scratch in float (192 B in the
    variant compiled for four wavefronts per SIMD), at most 256 B per lane in double (row-spectral D at 1024 pixels parks a few registers).  Those with the run-time-radix
    transform contain a butterfly for every radix; written carelessly, the prime-radix ones (one table read per term)
    made every such kernel keep 1 - 3 KB per lane in scratch whatever radices it ran.  Budgets: 16-byte-access kernels
    128 B in float (512 B for Chambolle-Pock's, which holds two rows, and with row-spectral D) and 1024 B in double,
    scalar-access kernels (compiled for 1024 threads: 128 registers) 1280 B."""
    import re
    fam = {n: k for n, k in kernels.items() if "k_solver_row" in n or "k_cp_row" in n}
    assert len(fam) >= 60, len(fam)
    seen = {"fixed": 0, "generic": 0}
    for n, k in fam.items():
        scratch = k[".private_segment_fixed_size"]
        double = n.startswith("_Z12k_solver_rowId") or n.startswith("_Z8k_cp_rowId")
        # mangled template arguments: ...ILi<V>ELi<NG>ELb<SPEC>E[Li<PF>E]Li<WF>ELi<MAXT>ELi<MINW>E
        ints = [int(v) for v in re.findall(r"Li(\d+)E", n)]
        wf, maxt = ints[-3], ints[-2]
        if wf != 0:
            seen["fixed"] += 1
            minw = ints[-1]
            if (minw == 4 or maxt == 1024) and not double:
                # the variants for images with several rounds of row workgroups per CU, and for rows of 4097 ... 8192
                # pixels on up to 1024 threads: four wavefronts per SIMD, 128
                # registers, a few of them parked (measured faster all the same: twice the workgroups resident)
                assert scratch <= 192 and k[".vgpr_count"] <= 128, (n, scratch, k[".vgpr_count"])
                continue
            wide_cp = "k_cp_row" in n and wf > 4096       # two rows of 4097 ... 8192 pixels beside radix-16 butterflies
            assert scratch <= (256 if double or wide_cp else 32), (n, scratch)
            assert k[".vgpr_count"] <= 256, (n, k[".vgpr_count"])
        else:
            seen["generic"] += 1
            limit = 1280 if maxt == 1024 else (1024 if double else 128)
            spec = "Lb1E" in n
            if (spec or "k_cp_row" in n) and maxt != 1024 and not double:
                limit = 512      # a second transform pair per row (row-spectral D) / two rows per workgroup
            assert scratch <= limit, (n, scratch, limit)
    assert seen["fixed"] >= 40 and seen["generic"] >= 16, seen
