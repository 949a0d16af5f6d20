"""Post-compile check of the gfx950 ISA of every kernel in the library (run by libhuffman_amd.build).

Why this exists (round 3): on MI355X a 64-bit VALU shift

    v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64   vdst[2], v<N>, vsrc[2]

takes its shift amount from the wrong place when v<N> is the LAST VGPR the wave has been allocated
(allocations are multiples of 8 registers; a kernel that uses exactly 8k registers and whose register
allocator put a shift amount into v<8k-1>).  `tools/calib/last_vgpr_probe.hip` reproduces it in sixty
lines for 24 ... 128 registers, for every wave that is not the first on its SIMD, and shows that one more
allocated register removes it.  hipcc 7.2 does not know the hazard: built for seven waves per SIMD
(72 VGPRs) `pack_kernel` kept the sixth code of every lane in v71 and shifted the lane's accumulator by
the LANE NUMBER instead of the code length - the "timing dependent" wrong payload bits of round 2.

The check: for every kernel whose VGPR count is a multiple of 8, no function of the module may use the
kernel's top register as the shift amount of a 64-bit shift.  A build that violates it fails loudly;
the remedy is one register of slack for that kernel (an `asm volatile("" ::: "v<8k>")` clobber, or a
different __launch_bounds__).
"""
from __future__ import annotations

import re

SHIFT64 = ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")
VGPR_GRANULE = 8

_KERNEL_RE = re.compile(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", re.S)
_LABEL_RE = re.compile(r"^([A-Za-z_][\w.$]*):")


def _functions(asm: str) -> dict:
    """label -> body lines, for every global (non .L) label that starts a function"""
    funcs, cur = {}, None
    for line in asm.split("\n"):
        m = _LABEL_RE.match(line)
        if m and not line.startswith(".L"):
            cur = m.group(1)
            funcs[cur] = []
        elif line.startswith(".L") and "func_begin" in line:
            continue
        elif cur is not None:
            funcs[cur].append(line)
    return funcs


def kernel_resources(asm: str) -> dict:
    """kernel symbol -> {vgprs, allocated, sgprs, lds, scratch} from the .amdhsa_kernel blocks"""
    res = {}
    for m in _KERNEL_RE.finditer(asm):
        name, body = m.group(1), m.group(2)

        def field(key, default=0):
            f = re.search(r"\.amdhsa_%s (\d+)" % key, body)
            return int(f.group(1)) if f else default
        n = field("next_free_vgpr")
        res[name] = {
            "vgprs": n,
            "allocated": -(-n // VGPR_GRANULE) * VGPR_GRANULE,
            "sgprs": field("next_free_sgpr"),
            "lds": field("group_segment_fixed_size"),
            "scratch": field("private_segment_fixed_size"),
            # gfx90a and later: architectural VGPRs in front of the AGPRs of the unified file (a multiple of 4;
            # equal to the VGPR count, rounded up, when the kernel has no AGPRs)
            "accum_offset": field("accum_offset", 0),
        }
    return res


def last_vgpr_shift_hazards(asm: str) -> list:
    """[(kernel, top register, [offending instructions])] - empty when the module is safe"""
    funcs = _functions(asm)
    kernels = kernel_resources(asm)
    # bodies of everything that is not itself a kernel: device functions a kernel may call
    callee_lines = [ln for name, body in funcs.items() if name not in kernels for ln in body]
    out = []
    for name, r in kernels.items():
        if r["vgprs"] == 0:
            continue
        tops = []
        if r["vgprs"] == r["allocated"]:
            tops.append(r["allocated"] - 1)     # (otherwise: at least one allocated register above the highest used one)
        # A kernel with AGPRs (accum_offset < next_free_vgpr): the probe was run on kernels without them, so which
        # register the hardware takes for "the last one" there is not known - the top ARCHITECTURAL register is
        # checked as well.
        acc = r.get("accum_offset", 0)
        if acc and acc < r["vgprs"] and acc - 1 not in tops:
            tops.append(acc - 1)
        for top in tops:
            pat = re.compile(r"^\s*(%s)(_e64)?\s+v\[\d+:\d+\],\s*v%d\s*," % ("|".join(SHIFT64), top))
            hits = [ln.strip() for ln in funcs.get(name, []) + callee_lines if pat.match(ln)]
            if hits:
                out.append((name, "v%d" % top, hits))
    return out


def format_hazards(hazards: list) -> str:
    lines = ["gfx950 last-VGPR shift hazard (libhuffman_amd/isa_check.py):"]
    for name, top, hits in hazards:
        lines.append(f"  kernel {name}: uses all of its VGPR allocation and {top} is the shift amount of")
        lines += [f"      {h}" for h in hits[:4]]
        if len(hits) > 4:
            lines.append(f"      ... {len(hits) - 4} more")
    lines.append("  give the kernel one register of slack (asm volatile(\"\" ::: \"v<allocation>\")) or change its __launch_bounds__")
    return "\n".join(lines)
