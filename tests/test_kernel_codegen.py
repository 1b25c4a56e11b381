"""Static checks of the tiled accumulate's generated code (no GPU needed: hipcc cross-compiles).

acc_tiled_kernel keeps its 64 FP64 accumulator pairs in v[128:255] and the working set of its generated
chunk loop (gen_acc_tiled.py: stream ring, prepared sets, LDS addresses, factor quads) in v[64:127], all
outside hipcc's register allocation; the ring's loads stay in flight across compiler code.  That is only
sound while the compiler itself never allocates a register at or above v64 and keeps scratch out of the
tile loop: both are properties of a particular hipcc, so they are asserted on the assembly this toolchain
emits."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "singlet_amd", "csrc")
SRC = os.path.join(CSRC, "kernels_tiled.hip")
HIPCC = "/opt/rocm/bin/hipcc"


def _device_asm(src, tmpdir):
    """Device assembly of a translation unit of the library.  The build keeps it as a by-product of the compilation that made
    the shipped object (singlet_amd/csrc/asm/<unit>.s, Makefile: -save-temps): use that when it is at least as new as every
    source it depends on -- it then describes exactly the code in libsinglet_hip.so -- else compile the unit here."""
    unit = os.path.splitext(os.path.basename(src))[0]
    kept = os.path.join(CSRC, "asm", unit + ".s")
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + [os.path.join(ROOT, "include", "singlet_hip.h")]
    gen = {"kernels_tiled": "acc_tiled_gen.inc", "kernels_nnls_asm": "nnls_lane_gen.inc", "kernels_nnls_half_asm": "nnls_half_gen.inc"}
    deps = [d for d in deps if not d.endswith("_gen.inc") or os.path.basename(d) == gen.get(unit)]   # a unit's own generated code only
    if unit == "kernels_tiled":
        deps.append(os.path.join(CSRC, "gen_acc_tiled.py"))
    if os.path.exists(kept) and all(os.path.getmtime(kept) >= os.path.getmtime(d) for d in deps):
        return open(kept).read()
    out = os.path.join(str(tmpdir), unit + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S",
                    "--cuda-device-only", "-o", out, src], check=True, capture_output=True, timeout=900)
    return open(out).read()


@pytest.fixture(scope="module")
def tiled_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    subprocess.run(["make", "-C", os.path.dirname(SRC), "acc_tiled_gen.inc"], check=True, capture_output=True, timeout=120)
    text = _device_asm(SRC, tmp_path_factory.mktemp("asm"))
    inst = {}
    for nsl in (2, 3, 4, 6, 7):   # acc_tiled_kernel<MODE>: pairs with prepared sets, pairs on the ring layout, quads (k <= 32); 6 / 7: 3 / 4 with the schedule table
        m = re.search(r"^(_Z16acc_tiled_kernelILi%dEE\w*):[^\n]*\n(.*?)s_endpgm" % nsl, text, re.S | re.M)
        assert m, "acc_tiled_kernel<%d> not found in the assembly" % nsl
        meta = text[text.index(".amdhsa_kernel " + m.group(1)):]
        inst[nsl] = (m.group(2), meta[:meta.index(".end_amdhsa_kernel")])
    return inst


def _vregs(line):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", line):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", line))
    return regs


@pytest.mark.parametrize("nsl", [2, 3, 4, 6, 7])
def test_compiler_stays_below_v64_and_keeps_scratch_out_of_the_loop(tiled_asm, nsl):
    body, meta = tiled_asm[nsl]
    in_asm, worst = False, -1
    for line in body.splitlines():
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        code = line.split(";")[0]
        if in_asm or not code.strip():
            continue
        r = _vregs(code)
        if r:
            worst = max(worst, max(r))
    assert 0 <= worst < 64, "hipcc allocated v%d: the asm-owned registers v[64:255] would be clobbered" % worst
    assert re.search(r"\.amdhsa_next_free_vgpr 256\b", meta), "the kernel descriptor must allocate all 256 VGPRs"
    # a few loop-invariant values may be parked in scratch around the tile loop; the chunk loop itself is asm
    m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta)
    assert m and int(m.group(1)) <= 64, "acc_tiled_kernel spills %s bytes per lane" % (m.group(1) if m else "?")


def test_chunk_loop_is_the_generated_asm_with_counted_waits(tiled_asm):
    """The chunk loop must be the generated block: four ring-slot bodies + four prologues, each preparing a set behind
    the counted stream wait vmcnt(6); one counted LDS wait per group of four entry pairs (a full drain only in front
    of the last group of a chunk), no vmcnt(0); vector destinations are only written while M0 indexes nothing."""
    body, _ = tiled_asm[2]
    blocks = re.findall(r"#ASMSTART(.*?)#ASMEND", body, re.S)
    chunk = [b for b in blocks if "v_fmac_f64_dpp" in b]
    assert len(chunk) == 1, "expected exactly one asm block with the FMAs"
    text = chunk[0]
    assert text.count("s_waitcnt vmcnt(6)") == 8          # 4 prologues + 4 in-loop preparations
    assert "vmcnt(0)" not in text
    assert text.count("v_fmac_f64_dpp") == 4 * (4 + 1) * 16  # 4 bodies x (4 octets + the no-prefetch copy of the last) x 16
    assert text.count("ds_read_b128") == 4 * 8 + 4 * 4 * 8  # prologues + one read behind every prefetching pair
    assert text.count("s_waitcnt lgkmcnt(4)") == 4 * (4 * 2 + 1) and text.count("s_waitcnt lgkmcnt(0)") == 4
    assert "v_readlane" not in text                       # pair switches are scalar (byte queue in SGPRs)
    # walk the block in program order: the cold section (last octets, pair switches) sits behind the loop and is
    # entered / left by branches, so the linear M0 state only holds for the hot part
    hot = text[:text.index("s_set_gpr_idx_off")]
    m0 = None
    for l in (x.strip() for x in hot.splitlines()):
        if l.startswith("s_mov_b32 m0"):
            m0 = l.split(",")[1].strip()
        elif l.startswith(".Ltiled_last") or l.startswith(".Ltiled_sw"):
            m0 = "cold"    # out-of-line code: FMAs behind a group head, no vector-destination VALU besides them
        elif l.startswith(("v_add_u32_dpp", "v_mov_b32", "v_permlane16_swap")):
            assert m0 == "0", "VALU with a vector destination while M0 indexes destinations: %s" % l


@pytest.mark.parametrize("mode", [4, 3])
def test_quad_chunk_loop_reads_its_operands_from_the_ring(tiled_asm, mode):
    """acc_tiled_kernel<4> (ranks up to 32, four columns per LDS instruction): eight ring-slot bodies of two octets, no
    set preparation (no v_mov / v_permlane16_swap: the DPP operands are the ring registers, written by vector loads
    only -- no VALU-write -> DPP-read hazard can exist), counted stream waits, destinations written only with M0 off."""
    body, _ = tiled_asm[mode]   # (mode 3: the pair layout on the same loop, half-set ring slots with doubled lane rows)
    blocks = re.findall(r"#ASMSTART(.*?)#ASMEND", body, re.S)
    chunk = [b for b in blocks if "v_fmac_f64_dpp" in b]
    assert len(chunk) == 1
    text = chunk[0]
    assert "v_permlane16_swap" not in text and "v_mov_b32 v" not in text and "v_readlane" not in text
    assert "vmcnt(0)" not in text
    assert text.count("s_waitcnt vmcnt(14)") == 8 and text.count("s_waitcnt vmcnt(12)") == 8
    assert text.count("v_fmac_f64_dpp") == 8 * (2 + 1) * 16
    assert text.count("ds_read_b128") == 8 * 8 + 8 * 2 * 8
    assert text.count("global_load_dword ") == 8 and text.count("global_load_dwordx2") == 8
    # DPP sources: ring registers only (v64..v71 row offsets, v[72:87] values)
    for l in (x.strip() for x in text.splitlines()):
        if l.startswith("v_add_u32_dpp"):
            src = int(re.match(r"v_add_u32_dpp v\d+, v(\d+),", l).group(1))
            assert 64 <= src <= 71, l
        elif l.startswith("v_fmac_f64_dpp"):
            src = int(re.match(r"v_fmac_f64_dpp v\[\d+:\d+\], v\[(\d+):\d+\],", l).group(1))
            assert 72 <= src <= 86 and src % 2 == 0, l
    hot = text[:text.index("s_set_gpr_idx_off")]
    m0 = None
    for l in (x.strip() for x in hot.splitlines()):
        if l.startswith("s_mov_b32 m0"):
            m0 = l.split(",")[1].strip()
        elif l.startswith(".Ltiled_last") or l.startswith(".Ltiled_sw"):
            m0 = "cold"
        elif l.startswith("v_add_u32_dpp"):
            assert m0 == "0", "VALU with a vector destination while M0 indexes destinations: %s" % l


@pytest.mark.parametrize("mode", [6, 7])
def test_schedule_table_chunk_loop(tiled_asm, mode):
    """acc_tiled_kernel<6 / 7> (the default): one s_bfe_u32 m0 per group of four entry tuples out of the 32 schedule words of
    the running lap, no countdown, no switch code, no branch inside an octet; the next lap's words loaded a lap ahead and
    waited with a full lgkmcnt drain (SMEM returns out of order); operands straight from the ring."""
    body, _ = tiled_asm[mode]
    blocks = re.findall(r"#ASMSTART(.*?)#ASMEND", body, re.S)
    chunk = [b for b in blocks if "v_fmac_f64_dpp" in b]
    assert len(chunk) == 1
    text = chunk[0]
    assert "v_permlane16_swap" not in text and "v_mov_b32 v" not in text and "v_readlane" not in text
    assert "vmcnt(0)" not in text
    assert text.count("v_fmac_f64_dpp") == 8 * (2 + 2) * 16      # eight bodies + eight copies for a chunk's last half-set
    heads = re.findall(r"s_bfe_u32 m0, s(\d+), (0x[0-9a-f]+)", text)
    assert len(heads) == 8 * (2 + 2) * 2                     # one per group; the last half-sets' copies included
    hot = [(int(r), int(f, 16)) for r, f in heads[:0]]
    seen = {(int(r), int(f, 16)) for r, f in heads}
    assert seen == {(52 + g // 2, (16 * (g & 1)) | (16 << 16)) for g in range(32)}   # every group of the lap, static
    assert text.count("s_load_dwordx16") == 3                # chunk entry (2) + the wrap of the ring (1)
    assert "s_sub_u32 s89" not in text and ".Ltiled_sw" not in text   # no group countdown, no switch code
    # between a group head and the end of its four tuples nothing but the wait, the FMAs and their reads
    for m in re.finditer(r"s_bfe_u32 m0[^\n]*\n(.*?)(?=s_bfe_u32 m0|s_mov_b32 m0, 0|s_branch|\Z)", text, re.S):
        for l in (x.strip().strip('"').strip() for x in m.group(1).splitlines()):
            if l and not l.endswith(":"):
                assert l.startswith(("s_waitcnt lgkmcnt", "v_fmac_f64_dpp", "ds_read_b128", "global_load", "s_add", "s_addc", "s_sub_u32 s88", "s_cmp", "s_cbranch", "s_waitcnt vmcnt", "s_mov_b64 s[", "s_load_dwordx16", "s_mov_b32 %", "s_mov_b32 s")), l
    hot_txt = text[:text.index("s_set_gpr_idx_off")]
    m0 = None
    for l in (x.strip() for x in hot_txt.splitlines()):
        if l.startswith("s_mov_b32 m0"):
            m0 = "0"
        elif l.startswith("s_bfe_u32 m0"):
            m0 = "idx"
        elif l.startswith("v_add_u32_dpp"):
            assert m0 == "0", "address add while M0 indexes destinations: %s" % l


def _dst_src0(code):
    ops = code.split(None, 1)[1] if " " in code or "\t" in code else ""
    parts = [p.strip() for p in ops.split(",")]
    return (parts[0].split()[0] if parts and parts[0] else ""), (parts[1].split()[0] if len(parts) > 1 and parts[1] else "")


def test_nnls_dpp_operands_have_no_valu_write_hazard(tmp_path):
    """nnls_lane_kernel<KP, true> feeds the Gram to its FMAs with inline-asm DPP instructions.  A DPP read
    of a VGPR needs two wait states after a VALU write of it, and hipcc pads nothing around inline asm: the
    DPP source registers must come straight from vector loads.  Also: no scratch."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path / "nnls50.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "singlet_amd", "csrc"), "-S", "--cuda-device-only", "-o", out,
                    os.path.join(ROOT, "tests", "codegen", "nnls_lane_50.hip")], check=True, capture_output=True, timeout=600)
    text = open(out).read()
    m = re.search(r"^_Z16nnls_lane_kernelILi50ELb1ELb0EE[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
    assert m, "kernel not found"
    prev, n_dpp = [], 0
    for line in m.group(1).splitlines():
        code = line.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        if "_dpp" in code:
            n_dpp += 1
            src0 = _vregs(_dst_src0(code)[1])
            for pl in prev[-2:]:
                if pl.startswith("v_") and (_vregs(_dst_src0(pl)[0]) & src0):
                    raise AssertionError("DPP hazard: %r followed by %r" % (pl, code))
        prev.append(code)
    assert n_dpp >= 50 * 50, "expected one DPP FMA per (coordinate, row entry)"
    d = text.index(".amdhsa_kernel _Z16nnls_lane_kernelILi50ELb1ELb0EE")
    ms = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", text[d:d + 1500])
    assert ms and int(ms.group(1)) <= 64, "NNLS lane kernel <50> spills %s bytes per lane" % (ms.group(1) if ms else "?")


def test_list_downdate_valu_remainder_rows_have_no_dpp_hazard(mask_asm):
    """mask_gram_list_kernel<NT, 1, 0, 0, REMV> (round 6): the rows beyond 16 NT are v_fmac_f64_dpp on the group's operand of block
    NT, which must reach the FMAs straight from its load -- a VALU write of a DPP source needs two wait states hipcc does not pad
    inside inline asm.  Every instance: REMV (NT + 1) DPP FMAs per group in the steady-state loop, none within two instructions of a
    VALU write of its broadcast source, no scratch, at most 256 registers (two workgroups per CU)."""
    found = 0
    for m in re.finditer(r"^(_Z21mask_gram_list_kernelILi(\d+)ELi1ELi0ELi0ELi(\d+)EE\w*):[^\n]*\n(.*?)s_endpgm", mask_asm, re.S | re.M):
        name, nt, remv, body = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)
        if remv == 0:
            continue
        found += 1
        prev, n_dpp = [], 0
        for line in body.splitlines():
            code = line.split(";")[0].strip()
            if not code or code.startswith(".") or code.endswith(":"):
                continue
            if code.startswith("v_fmac_f64_dpp"):
                n_dpp += 1
                src0 = _vregs(_dst_src0(code)[1])
                for pl in prev[-2:]:
                    if pl.startswith("v_") and (_vregs(_dst_src0(pl)[0]) & src0):
                        raise AssertionError("NT=%d REMV=%d DPP hazard: %r followed by %r" % (nt, remv, pl, code))
            prev.append(code)
        assert n_dpp >= 3 * remv * (nt + 1), (nt, remv, n_dpp)      # three groups per lap of the steady-state loop (+ the tails)
        meta = mask_asm[mask_asm.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1)) == 0, (nt, remv)
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)) <= 256, (nt, remv)
    assert found == 7, found


def test_four_lanes_per_column_solve_has_no_dpp_hazard_and_no_scratch(tmp_path_factory):
    """nnls_quarter_kernel<KQ> (nnls_quarter.h, instances in kernels_nnls_quarter1 / 2.hip; ranks 129 - 256): the row update is inline-asm v_fmac_f64_dpp on Gram
    pieces that must reach the FMAs straight from their loads (or a copy at least two instructions old): hipcc pads nothing
    around inline asm.  Every instance: no VALU write of a DPP source within two instructions of its read, one DPP FMA per
    (coordinate, entry of the lane's quarter), no scratch (the larger instances overflow into AGPRs, not memory)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    text = "".join(_device_asm(os.path.join(CSRC, "kernels_nnls_quarter%d.hip" % part), tmp_path_factory.mktemp("asm")) for part in (1, 2))
    for kq in (36, 40, 44, 48, 52, 56, 60, 64):
        m = re.search(r"^(_Z19nnls_quarter_kernelILi%dEE\w*):[^\n]*\n(.*?)s_endpgm" % kq, text, re.S | re.M)
        assert m, "nnls_quarter_kernel<%d> not found" % kq
        prev, n_dpp = [], 0
        for line in m.group(2).splitlines():
            code = line.split(";")[0].strip()
            if not code or code.startswith(".") or code.endswith(":"):
                continue
            if "_dpp" in code:
                n_dpp += 1
                src0 = _vregs(_dst_src0(code)[1])
                for pl in prev[-2:]:
                    if pl.startswith("v_") and (_vregs(_dst_src0(pl)[0]) & src0):
                        raise AssertionError("KQ=%d DPP hazard: %r followed by %r" % (kq, pl, code))
            prev.append(code)
        assert n_dpp >= 4 * kq * kq, (kq, n_dpp)
        meta = text[text.index(".amdhsa_kernel " + m.group(1)):]
        ms = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta[:meta.index(".end_amdhsa_kernel")])
        assert ms and int(ms.group(1)) == 0, (kq, ms.group(1) if ms else None)


# ---- mask_gram_list_kernel: accumulator tiles in named AGPRs outside hipcc's allocation ------------------------------
MASK_SRC = os.path.join(ROOT, "singlet_amd", "csrc", "kernels_mask.hip")


@pytest.fixture(scope="module")
def mask_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    return _device_asm(MASK_SRC, tmp_path_factory.mktemp("asm"))


def test_list_downdate_kernels_own_their_agprs(mask_asm):
    """mask_gram_list_kernel names its accumulators a[0 : NREG - 1] in inline asm.  That is only sound while hipcc
    itself never touches the AGPR file in these kernels (no spills to AGPRs, no MFMA of its own), does not spill to
    scratch, and the kernel descriptor covers the named registers: every AGPR reference must sit inside an asm
    statement, and the steady-state loop must not wait for vmcnt(0) before every group (branch-free loop)."""
    kernels = re.findall(r"^(_Z21mask_gram_list_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E\w*):[^\n]*\n(.*?)^\.Lfunc_end", mask_asm, re.S | re.M)
    assert len(kernels) >= 20, "list kernel instances not found"
    for name, nt, nparts, part, rem, body in kernels:
        nt, nparts, part, rem = int(nt), int(nparts), int(part), int(rem)
        ntiles = (nt * (nt + 1) // 2 - part + nparts - 1) // nparts
        nreg = 8 * ntiles + 2 * (nt + 1) * ((rem + 3) // 4) if rem else 8 * ntiles   # tiles + remainder quads x column blocks
        in_asm = False
        for line in body.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            code = line.split(";")[0]
            if not in_asm and re.search(r"\ba(\d+|\[)", code):
                raise AssertionError("%s: hipcc uses an AGPR itself: %s" % (name, code.strip()))
        assert "scratch_" not in body, "%s spills to scratch" % name
        asm_text = "\n".join(re.findall(r"#ASMSTART(.*?)#ASMEND", body, re.S))
        assert asm_text.count("v_accvgpr_write_b32") == nreg, name
        used = [int(x, 0) for x in re.findall(r"\ba\[(0x[0-9a-fA-F]+|\d+)", asm_text)] + [int(x, 0) for x in re.findall(r"\ba\[[0-9a-fx]+:(0x[0-9a-fA-F]+|\d+)\]", asm_text)]
        assert max(used) < nreg, name
        meta = mask_asm[mask_asm.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        acc_off = int(re.search(r"\.amdhsa_accum_offset (\d+)", meta).group(1))
        nxt = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1))
        assert nxt - acc_off >= nreg, "%s: the descriptor allocates %d AGPRs, the asm names %d" % (name, nxt - acc_off, nreg)
        # the pipelined loop: the block holding 3 groups of MFMAs must contain counted waits, not only vmcnt(0)
        blocks = re.split(r"^\.LBB\d+_\d+:", body, flags=re.M)
        best = max(blocks, key=lambda b: b.count("v_mfma_f64_16x16x4"))
        waits = re.findall(r"vmcnt\((\d+)\)", best)
        assert any(int(w) > 1 for w in waits), "%s: no counted vmcnt in the steady-state loop (%s)" % (name, waits)


# ---- nnls_half_kernel<KH > 52>: x in named AGPRs outside hipcc's allocation -------------------------------------------
HALF_SRC = os.path.join(ROOT, "singlet_amd", "csrc", "kernels_nnls_half.hip")


def test_half_lane_nnls_x_in_agprs(tmp_path_factory):
    """The k = 105 ... 128 instances keep x in a[0 : 2 KH - 1] by name: hipcc itself must not touch the AGPR file there
    (it would, to spill, if the kernel were held to 256 registers), must not spill to scratch, and the kernel
    descriptor must cover the named registers."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    text = _device_asm(HALF_SRC, tmp_path_factory.mktemp("asm"))
    kernels = re.findall(r"^(_Z16nnls_half_kernelILi(\d+)E\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    assert len(kernels) == 8
    for name, kh, body in kernels:
        kh = int(kh)
        in_asm = False
        for line in body.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            code = line.split(";")[0]
            if not in_asm and re.search(r"\ba(\d+|\[)", code):
                raise AssertionError("%s: hipcc uses an AGPR itself: %s" % (name, code.strip()))
        meta = text[text.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1))
        if kh > 52:
            assert scratch == 0, "%s spills %d bytes" % (name, scratch)
            acc_off = int(re.search(r"\.amdhsa_accum_offset (\d+)", meta).group(1))
            nxt = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1))
            assert nxt - acc_off >= 2 * kh, "%s: %d AGPRs allocated, %d named" % (name, nxt - acc_off, 2 * kh)
            asm_text = "\n".join(re.findall(r"#ASMSTART(.*?)#ASMEND", body, re.S))
            assert max(int(x, 0) for x in re.findall(r"\ba\[(0x[0-9a-fA-F]+|\d+)\]", asm_text)) == 2 * kh - 1
        else:
            assert scratch <= 64, "%s spills %d bytes" % (name, scratch)


# ---- nnls_lane_asm_kernel_<KP>: the whole solve of a column in ONE asm statement that owns v[V_T : TOP - 1] -----------------------
ASM_NNLS_SRC = os.path.join(CSRC, "kernels_nnls_asm.hip")


def test_generated_nnls_sweep_owns_its_registers(tmp_path_factory):
    """kernels_nnls_asm.hip keeps b, x and the sweep's working set in named registers v[V_T : TOP - 1] (gen_nnls_lane.py's plan),
    all clobbers of the one asm statement that holds a column's solve: hipcc must keep its own values (the statement's
    operands included) below V_T, the kernel descriptor must cover TOP registers and no more (the plan is what sets the waves
    per SIMD), there must be exactly one such statement per kernel with KP x KP row-update FMAs per sweep, and no scratch access
    inside it."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    subprocess.run(["make", "-C", CSRC, "nnls_lane_gen.inc"], check=True, capture_output=True, timeout=120)
    inc = open(os.path.join(CSRC, "nnls_lane_gen.inc")).read()
    kps = [int(v) for v in re.findall(r"X_\((\d+)\)", inc.split("SGL_NNLS_ASM_INSTANCES(X_)")[1].splitlines()[0])]
    assert kps and max(kps) <= 64
    text = _device_asm(ASM_NNLS_SRC, tmp_path_factory.mktemp("asm"))
    for KP in kps:
        vt = int(re.search(r"#define NNLS_ASM_VT_%d (\d+)" % KP, inc).group(1))
        clob = re.search(r"#define NNLS_ASM_VCLOB_%d (.*)" % KP, inc).group(1)
        top = max(int(v) for v in re.findall(r'"v(\d+)"', clob)) + 1
        nacc = len(re.findall(r'"a\d+"', clob))      # ranks above 50: x in a[0 : 2 KP - 1]
        assert nacc == (2 * KP if KP > 50 else 0)
        m = re.search(r"^(_Z\d+nnls_lane_asm_kernel_%d\w*):[^\n]*\n(.*?)s_endpgm" % KP, text, re.S | re.M)
        assert m, "nnls_lane_asm_kernel_%d not found" % KP
        body = m.group(2)
        meta = text[text.index(".amdhsa_kernel " + m.group(1)):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        in_asm, worst = False, -1
        for line in body.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            code = line.split(";")[0]
            if in_asm or not code.strip():
                continue
            r = _vregs(code)
            if r:
                worst = max(worst, max(r))
        assert 0 <= worst < vt, "KP=%d: hipcc allocated v%d, the generated sweep owns v[%d:%d]" % (KP, worst, vt, top - 1)
        nfree = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1))
        assert top + nacc <= nfree <= (top + 7) // 8 * 8 + nacc, (KP, nfree, top, nacc)
        blocks = [b for b in re.findall(r"#ASMSTART(.*?)#ASMEND", body, re.S) if "v_fmac_f64_dpp" in b]
        assert len(blocks) == 1
        assert blocks[0].count("v_fmac_f64_dpp") == KP * KP
        assert "scratch_" not in blocks[0] and "buffer_" not in blocks[0]
        # every transcendental result is at least one instruction away from its first reader (gfx940+ hazard)
        lines = [x.strip() for x in blocks[0].splitlines() if x.strip()]
        for a, b in zip(lines, lines[1:]):
            mm = re.match(r"v_rcp_f64 (v\[\d+:\d+\])", a)
            if mm:
                assert mm.group(1) not in b.split(",", 1)[-1] or b.startswith("s_nop"), (a, b)


ASM_NNLS_HALF_SRC = os.path.join(CSRC, "kernels_nnls_half_asm.hip")


def test_generated_two_lane_solve_owns_its_registers(tmp_path_factory):
    """kernels_nnls_half_asm.hip (ranks 65 ... 128, gen_nnls_half.py): the kernel descriptor covers the register plan and leaves the
    waves per SIMD the plan counts on, one asm statement per kernel with KP x KH row-update FMAs per sweep, every register it
    names inside the plan or an operand hipcc placed outside it, no scratch access inside it -- and the two hazards the generator pads itself: a transcendental's result is not read by the next
    instruction, a v_permlane32_swap does not read a register a VALU instruction wrote within the two instructions before."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    subprocess.run(["make", "-C", CSRC, "nnls_half_gen.inc"], check=True, capture_output=True, timeout=120)
    inc = open(os.path.join(CSRC, "nnls_half_gen.inc")).read()
    kps = [int(v) for v in re.findall(r"X_\((\d+)\)", inc.split("SGL_NNLS_HALF_ASM_INSTANCES(X_)")[1].splitlines()[0])]
    assert kps and min(kps) >= 68 and max(kps) <= 128
    text = _device_asm(ASM_NNLS_HALF_SRC, tmp_path_factory.mktemp("asm"))
    for KP in kps:
        KH = KP // 2
        vt = int(re.search(r"#define NNLS_HALF_ASM_VT_%d (\d+)" % KP, inc).group(1))
        clob = re.search(r"#define NNLS_HALF_ASM_VCLOB_%d (.*)" % KP, inc).group(1)
        top = max(int(v) for v in re.findall(r'"v(\d+)"', clob)) + 1
        nacc = len(re.findall(r'"a\d+"', clob))      # ranks above 100: x in a[0 : 2 KH - 1]
        assert nacc == (2 * KH if KP > 100 else 0)
        m = re.search(r"^(_Z\d+nnls_half_asm_kernel_%d\w*):[^\n]*\n(.*?)s_endpgm" % KP, text, re.S | re.M)
        assert m, "nnls_half_asm_kernel_%d not found" % KP
        body = m.group(2)
        meta = text[text.index(".amdhsa_kernel " + m.group(1)):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        # (the solve is ONE statement and its registers are its clobbers: hipcc cannot keep a live value in them across it, and is
        #  free to use them as temporaries before and after -- it does; what has to hold is that the descriptor covers the plan)
        waves = int(re.search(r"#define NNLS_HALF_ASM_WAVES_%d (\d+)" % KP, inc).group(1))
        nfree = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1))
        assert top + nacc <= nfree and 512 // ((nfree + 7) // 8 * 8) >= waves, (KP, nfree, top, nacc, waves)
        blocks = [b for b in re.findall(r"#ASMSTART(.*?)#ASMEND", body, re.S) if "v_fmac_f64_dpp" in b]
        assert len(blocks) == 1
        assert blocks[0].count("v_fmac_f64_dpp") == KP * KH
        assert blocks[0].count("v_permlane32_swap_b32") == 2 * KP + 4      # nd of every coordinate, tol twice per sweep
        assert "scratch_" not in blocks[0] and "buffer_" not in blocks[0]
        lines = [x.strip() for x in blocks[0].splitlines() if x.strip()]
        named = set()
        for ins in lines:
            named |= _vregs(ins.split(";")[0])
        outside = {v for v in named if not (vt <= v < top)}
        assert len(outside) <= 16 and max(named) < nfree, (KP, sorted(outside))   # the statement's vector operands (eleven registers, hipcc may pass a pair twice)
        for a, b in zip(lines, lines[1:]):
            mm = re.match(r"v_rcp_f64 (v\[\d+:\d+\])", a)
            if mm:
                assert mm.group(1) not in b.split(",", 1)[-1] or b.startswith("s_nop"), (a, b)
        for q, ins in enumerate(lines):
            if not ins.startswith("v_permlane32_swap_b32"):
                continue
            regs = {int(v) for v in re.findall(r"v(\d+)", ins)}
            wait = 0
            for prev in reversed(lines[max(0, q - 2):q]):   # the two instructions before: an s_nop N counts N + 1 wait states
                if prev.startswith("s_nop"):
                    wait += int(prev.split()[1]) + 1
                    continue
                if wait >= 2:
                    break
                if prev.startswith("v_") and not prev.startswith("v_cmp"):
                    d = prev.split(None, 1)[1].split(",")[0].strip()
                    lo, hi = (d[2:-1].split(":") if d.startswith("v[") else (d[1:], d[1:]))
                    assert not (regs & set(range(int(lo), int(hi) + 1))), (KP, prev, ins)
                wait += 1
