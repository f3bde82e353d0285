#!/usr/bin/env python3
"""The instruction mix of the build kernels, from their ISA -- so that the index build's issue ceiling comes from the code
and from measured per-opcode costs, not from a kernel's own run time.

Compiles miekki_amd/csrc/build.hip for gfx950 with the library's flags to assembly (`hipcc --cuda-device-only -S`: the
same code generation as the shipped object), takes the text of the build kernels, counts the opcodes, and prices the
vector-ALU ones with the costs tools/ubench measured on the chip (profiles/r4_ubench.txt, cycles per wave-instruction
and SIMD at 8 waves per SIMD; an opcode that was not measured takes the price of its class, and the share that was is
reported).  The mix is STATIC -- the kernel's text, where the fully unrolled hash loop is most of the scatter kernel; the
DYNAMIC instruction counts come from the SQ counters (profiles/pmc_build.json, tools/pmc_build.py).  No GPU needed.

    python tools/isa_mix.py [--ubench profiles/r4_ubench.txt] > profiles/isa_mix.json
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "miekki_amd", "csrc", "build.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include")]
KERNELS = {
    "build_scatter_kernel<1,true>": r"_ZN2mk20build_scatter_kernelILi1ELb1E",
    "build_scatter_kernel<2,true>": r"_ZN2mk20build_scatter_kernelILi2ELb1E",
    "build_reduce_kernel<1,true>": r"_ZN2mk19build_reduce_kernelILi1ELb1E",
    "build_reduce_kernel<2,false>": r"_ZN2mk19build_reduce_kernelILi2ELb0E",
}
# opcode -> the ubench line that prices it (same functional unit, same operand width)
CLASS = [
    (r"v_(add|subrev)(_co)?_u32|v_(add|sub|subrev)_nc_u32|v_addc_co_u32|v_subb_co_u32|v_sub_co_u32|v_not_b32|v_accvgpr\w*", "v_add_u32"),
    (r"v_bitop3_b32", "v_bitop3_b32"),
    (r"v_ashrrev_i32|v_bfe_i32|v_bfi_b32|v_perm_b32", "v_bfe_u32"),
    (r"v_mbcnt_hi_u32_b32|v_bcnt_u32_b32", "v_mbcnt_lo_u32_b32"),
    (r"v_max_u32|v_min3_u32|v_max3_u32|v_min_i32|v_max_i32", "v_min_u32"),
    (r"v_(lshlrev|lshrrev|ashrrev)_b64", "v_lshlrev_b64"),
    (r"v_alignbit_b32|v_alignbyte_b32", "v_alignbit_b32"),
    (r"v_mul_lo_u32", "v_mul_lo_u32"), (r"v_mul_hi_u32", "v_mul_hi_u32"), (r"v_mad_u64_u32", "v_mad_u64_u32"),
    (r"v_mul_u32_u24|v_mul_hi_u32_u24", "v_mul_u32_u24"), (r"v_mad_u32_u24", "v_mad_u32_u24"),
    (r"v_add_lshl_u32", "v_lshl_add_u32"),
    (r"v_ffbh_u32|v_ffbl_b32|v_bcnt_u32_b32", "v_ffbh_u32"),
    (r"v_cmp_\w+_(u|i)64|v_cmpx_\w+_(u|i)64", "v_cmp_lt_u64+cndmask"),
    (r"v_cmp_\w+|v_cmpx_\w+", "v_cmp_ne_u32"),
]
INDIVIDUAL = {"v_xor3_b32": "v_or3_b32", "v_xad_u32": "v_and_or_b32", "v_cndmask_b32": "v_cmp_lt_u64+cndmask"}


def ubench_costs(path):
    """{op: cycles per wave-instruction per SIMD at 8 waves per SIMD} from tools/ubench's output."""
    costs = {}
    for line in open(path):
        m = re.match(r"^(v_\S+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line)
        if m:
            costs[m.group(1)] = float(m.group(5))
    return costs


def price(op, costs):
    if op in INDIVIDUAL and INDIVIDUAL[op] in costs:
        return costs[INDIVIDUAL[op]], op in costs or INDIVIDUAL[op] == op
    if op in costs:
        return costs[op], True
    for pat, ref in CLASS:
        if ref and re.fullmatch(pat, op) and ref in costs:
            return costs[ref], ref == op
    return None, False


def main():
    ub = os.path.join(ROOT, "profiles", "r4_ubench.txt")
    if "--ubench" in sys.argv:
        ub = sys.argv[sys.argv.index("--ubench") + 1]
    if not os.path.exists(ub):
        ub = os.path.join(ROOT, "profiles", "r3_ubench.txt")
    costs = ubench_costs(ub)
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "build.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "-S", "-o", asm, SRC], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read().splitlines()
    out = {"source": "miekki_amd/csrc/build.hip", "build_hip_sha256": hashlib.sha256(open(SRC, "rb").read()).hexdigest(),
           "flags": " ".join(f for f in FLAGS if not f.startswith("-I")), "ubench": os.path.relpath(ub, ROOT),
           "note": "static opcode counts of each kernel's text; weighted_cycles_per_valu = sum(count x measured cost) / sum(count) over the "
                   "vector-ALU opcodes, costs = cycles per wave-instruction and SIMD at 8 waves per SIMD (tools/ubench); the guide's peak is 2 "
                   "(MI355X_MICROARCH.md, Wave scheduling)", "kernels": {}}
    for name, pat in KERNELS.items():
        start = next((i for i, l in enumerate(text) if re.match(pat + r".*:\s", l + " ")), None)
        if start is None:
            continue
        end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
        ops = Counter()
        for l in text[start + 1:end]:
            m = re.match(r"^\s+([a-z_0-9]+)\b", l)
            if m and not m.group(1).startswith("."):
                ops[re.sub(r"_(e32|e64|sdwa|dpp)$", "", m.group(1))] += 1       # (encodings of one operation)
        valu = {o: n for o, n in ops.items() if o.startswith("v_") and not re.match(r"v_(readlane|readfirstlane|writelane|nop|mfma)", o)}
        tot = sum(valu.values())
        wsum = measured = fallback_n = 0.0
        unpriced = {}
        default = sorted(costs.values())[len(costs) // 2] if costs else 4.3
        for o, n in valu.items():
            c, exact = price(o, costs)
            if c is None:
                c = default
                unpriced[o] = n
            wsum += c * n
            measured += n if exact else 0
        out["kernels"][name] = {
            "instructions": sum(ops.values()), "valu": tot, "salu": sum(n for o, n in ops.items() if o.startswith("s_")),
            "lds": sum(n for o, n in ops.items() if o.startswith("ds_")),
            "vmem": sum(n for o, n in ops.items() if re.match(r"(global|buffer|flat|scratch)_", o)),
            "weighted_cycles_per_valu": wsum / max(tot, 1), "share_priced_by_its_own_measurement": measured / max(tot, 1),
            "share_priced_by_class_median": sum(unpriced.values()) / max(tot, 1),
            "valu_by_opcode": dict(sorted(valu.items(), key=lambda kv: -kv[1])[:24]),
            "lds_by_opcode": {o: n for o, n in ops.items() if o.startswith("ds_")},
        }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
