"""The build-time guard against the gfx950 last-VGPR shift hazard (libhuffman_amd/isa_check.py).

CPU-only: hipcc cross-compiles the kernels to gfx950 assembly here.  What the hazard is and how it was
found: DESIGN.md 3.3; the hardware reproducer is tools/calib/last_vgpr_probe.hip (run by
tests/test_gpu_pack_builds.py on the GPU box).
"""
import pytest

from libhuffman_amd import build, isa_check


def kernel_asm(name, vgprs, body, callee=""):
    return f"""
	.text
{callee}
	.globl	{name}
{name}:                                   ; @{name}
; %bb.0:
{body}
	s_endpgm
.Lfunc_end0:
	.section	.rodata
	.amdhsa_kernel {name}
		.amdhsa_group_segment_fixed_size 1024
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_next_free_vgpr {vgprs}
		.amdhsa_next_free_sgpr 20
	.end_amdhsa_kernel
"""


def test_shift_amount_in_the_last_allocated_vgpr_is_flagged():
    asm = kernel_asm("k72", 72, "\tv_lshlrev_b64 v[24:25], v71, v[24:25]\n\tv_lshrrev_b64 v[2:3], v70, v[24:25]")
    (name, top, hits), = isa_check.last_vgpr_shift_hazards(asm)
    assert (name, top) == ("k72", "v71") and hits == ["v_lshlrev_b64 v[24:25], v71, v[24:25]"]
    for op in isa_check.SHIFT64:
        assert isa_check.last_vgpr_shift_hazards(kernel_asm("k", 64, f"\t{op} v[0:1], v63, v[4:5]"))
    assert "v71" in isa_check.format_hazards([(name, top, hits)])


def test_one_register_of_slack_or_another_operand_is_not_flagged():
    shift = "\tv_lshlrev_b64 v[24:25], v71, v[24:25]"
    assert not isa_check.last_vgpr_shift_hazards(kernel_asm("k73", 73, shift))          # allocation 80
    assert not isa_check.last_vgpr_shift_hazards(kernel_asm("k71", 71, "\tv_lshlrev_b64 v[24:25], v70, v[24:25]"))
    # the VALUE may live in the last pair, 32-bit shifts and v_mad_u64_u32 are not affected (measured)
    ok = "\tv_lshlrev_b64 v[0:1], v3, v[70:71]\n\tv_lshlrev_b32_e32 v1, v71, v2\n\tv_mad_u64_u32 v[0:1], vcc, v71, v2, v[4:5]"
    assert not isa_check.last_vgpr_shift_hazards(kernel_asm("k72", 72, ok))


def test_a_called_function_counts_for_the_kernel():
    callee = "helper:                                 ; @helper\n\tv_ashrrev_i64 v[0:1], v39, v[2:3]\n\ts_setpc_b64 s[30:31]\n.Lfunc_end9:\n"
    assert isa_check.last_vgpr_shift_hazards(kernel_asm("k40", 40, "\ts_swappc_b64 s[30:31], s[4:5]", callee))
    assert not isa_check.last_vgpr_shift_hazards(kernel_asm("k48", 48, "\ts_swappc_b64 s[30:31], s[4:5]", callee))


def test_resources_are_read_from_the_kernel_descriptor():
    r = isa_check.kernel_resources(kernel_asm("k", 75, ""))["k"]
    assert r == {"vgprs": 75, "allocated": 80, "sgprs": 20, "lds": 1024, "scratch": 0, "accum_offset": 0}


def test_a_kernel_with_agprs_has_its_top_architectural_register_checked():
    """with AGPRs behind the architectural registers (accum_offset < next_free_vgpr) the guard must
    not depend on next_free_vgpr alone"""
    shift = "\tv_lshlrev_b64 v[24:25], v63, v[24:25]"
    with_agprs = kernel_asm("ka", 75, shift).replace(".amdhsa_next_free_sgpr 20", ".amdhsa_next_free_sgpr 20\n\t\t.amdhsa_accum_offset 64")
    (name, top, hits), = isa_check.last_vgpr_shift_hazards(with_agprs)
    assert (name, top) == ("ka", "v63")
    # no AGPRs (accum_offset at or above the VGPR count): only the last allocated register counts
    without = kernel_asm("kb", 75, shift).replace(".amdhsa_next_free_sgpr 20", ".amdhsa_next_free_sgpr 20\n\t\t.amdhsa_accum_offset 76")
    assert not isa_check.last_vgpr_shift_hazards(without)


@pytest.fixture(scope="module")
def shipped_table():
    return build.check_isa(extra_flags=[])


def test_the_shipped_kernels_are_clean(shipped_table):
    """the library as built by __graft_entry__.build(): every kernel present, none with the hazard"""
    names = " ".join(shipped_table)
    for k in ("hist_tree_kernel", "pack_kernel", "decode_kernel", "decode_sub_kernel", "decode_prepare_kernel"):
        assert k in names
    assert all(r["scratch"] == 0 for n, r in shipped_table.items() if "pack_kernelILi256ELb1" in n)
    # decode_sub: nothing spilled in its loops; the 16 bytes are registers saved around the out-of-line call of the
    # step-by-step path BEHIND the tile loop (dsub_tile_slow), which is next to never taken
    assert all(r["scratch"] <= 32 for n, r in shipped_table.items() if "decode_sub_kernel" in n)


def test_the_decoders_keep_four_workgroups_a_cu(shipped_table):
    """round 6: `__launch_bounds__(512, 8)` is a wish - the compiler drops to 7 waves a SIMD (three workgroups a CU) with a
    warning, and a kernel's register count is the largest of its own and its out-of-line callees': dec_build_tables, no longer
    inlined once a third kernel called it, took 70 registers and decode_fast_kernel / probe_kernel with it (1.15 -> 1.4 ms per
    GiB, unnoticed for a dozen commits).  The kernels whose design is four workgroups of 512 a CU: 64 registers, 39.6 KiB."""
    for k in ("decode_fast_kernel", "probe_kernelILi512", "decode_kernel", "spec_scan_kernel"):
        rows = [r for n, r in shipped_table.items() if k in n]
        assert rows, k
        for r in rows:
            assert r["allocated"] <= 64, (k, r)
            assert r["lds"] <= 40960, (k, r)
    assert all(r["allocated"] <= 64 for n, r in shipped_table.items() if "decode_sub_kernel" in n)
    assert all(r["allocated"] <= 80 for n, r in shipped_table.items() if "pack_kernelILi256ELb1" in n)


def test_the_seven_wave_pack_build_is_rejected_and_its_slack_build_accepted():
    """round 2's 'faster, and wrong' build: the 64-bit accumulator (-DPACK_ACC64), 72 of 72 VGPRs with code[5] in v71"""
    with pytest.raises(RuntimeError) as e:
        build.check_isa(extra_flags=["-DPACK_ACC64", "-DPACK_WAVES_PER_SIMD=7"])
    assert "pack_kernel" in str(e.value) and "v71" in str(e.value)
    table = build.check_isa(extra_flags=["-DPACK_ACC64", "-DPACK_WAVES_PER_SIMD=7", '-DPACK_VGPR_SLACK="v72"'])
    (vg,) = [r["vgprs"] for n, r in table.items() if "pack_kernelILi256ELb1" in n]
    assert vg == 73


def test_the_short_code_accumulator_has_no_64_bit_shift_left():
    """round 3: PackAcc<uint32_t> - the accumulator of the shipped short-code pack_kernel shifts 32 bits at a time
    (what is left of 64-bit shifts there: the header's block length by a byte count, addresses by constants), and
    the seven-wave build passes the check"""
    asm = build.device_asm([])
    body = asm[asm.index("_ZN6hufgpu11pack_kernelILi256ELb1EE"):]
    body = body[:body.index(".amdhsa_kernel")]
    assert "v_lshlrev_b64" not in body
    table = build.check_isa(extra_flags=["-DPACK_WAVES_PER_SIMD=7"])
    (vg,) = [r["vgprs"] for n, r in table.items() if "pack_kernelILi256ELb1" in n]
    assert vg <= 72
