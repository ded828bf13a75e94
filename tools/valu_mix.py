#!/usr/bin/env python3
"""Static vector-ALU instruction mix of the front-end kernels, by ISSUE-COST class, for the `valu_issue` roofline of bench.py.

The integer front end (k_fast_keypoints, k_select, k_brief, k_match) is bound by vector-instruction ISSUE, not by HBM.  On gfx950 a
wave-instruction does not cost the same whatever it is (profiles/r03_valu_issue_rates.txt, measured with tools/valu_rate.hip at >= 2
wavefronts per SIMD): plain 32-bit / 16-bit VOP1 / VOP2 arithmetic and logic issue every ~2.3 cycles, everything VOP3-only, packed,
compare / select, DPP, 64-bit or f64 every ~4.2, v_mad_u16 8.2, a v_cndmask fed by vcc ~16, the i8 matrix-core products 16.
This tool compiles the two files to ISA (hipcc -S, no GPU needed), classifies every vector instruction of a kernel's text and prints
per kernel the static counts per class and the mean issue cycles per vector instruction -- the weight bench.py multiplies the DYNAMIC
instruction count per wavefront with (SQ_INSTS_VALU / SQ_WAVES from the PMC pass, tools/pmc_front_end.sh).  The static mix stands in
for the dynamic one (loops execute their bodies in the listed proportion to first order); the result is a fraction of issue TIME.
    python tools/valu_mix.py [--json]
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNELS = {"orb_kernels.hip": ("k_fast_keypoints", "k_select", "k_brief"), "match_kernels.hip": ("k_match",)}

FAST = 2.3    # cycles per wave-instruction, >= 2 wavefronts per SIMD (r03_valu_issue_rates.txt, columns 2 / 4)
SLOW = 4.2
FAST_OPS = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
            "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_bitop3_b32", "v_min_u16", "v_max_u16", "v_min_i16", "v_max_i16",
            "v_add_u16", "v_sub_u16", "v_lshlrev_b16", "v_lshrrev_b16", "v_mul_lo_u16", "v_xnor_b32", "v_subrev_u16", "v_add_i32", "v_sub_i32"}


def cost(mn: str, operands: str) -> float:
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
    if base.startswith("v_mfma") or base.startswith("v_smfmac"):
        return 16.0
    if base == "v_mad_u16":
        return 8.2
    if base == "v_cndmask_b32":
        return 16.3 if "vcc" in operands else 4.2
    if mn.endswith("_dpp") or mn.endswith("_sdwa") or "row_" in operands or "quad_perm" in operands:
        return SLOW
    return FAST if base in FAST_OPS else SLOW


def kernel_bodies(asm: str):
    """name -> list of instruction lines of every kernel symbol of the file."""
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                cur = None
                continue
            s = line.strip()
            if s and not s.startswith((".", ";")) and not s.endswith(":"):
                cur.append(s)
    return out


def analyse():
    from mageslam_amd import build
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for src, names in KERNELS.items():
            path = os.path.join(ROOT, "mageslam_amd", "csrc", src)
            out = os.path.join(d, src + ".s")
            subprocess.check_call([build.HIPCC, *[f for f in build.flags_for(path) if f != "-fPIC"], "-S", "--cuda-device-only", path, "-o", out],
                                  stderr=subprocess.DEVNULL)
            bodies = kernel_bodies(open(out).read())
            for k in names:
                # of several instantiations the one the defaults run: k_fast_keypoints<2> = the Gaussian fused in, on the matrix cores
                cands = [n for n in bodies if re.search(rf"\d+{k}(I|E)", n)]
                cands.sort(key=lambda n: (0 if "ILi2E" in n else 1, n))
                if not cands:
                    continue
                counts = {"fast_2.3": 0, "slow_4.2": 0, "other": 0, "mfma": 0}
                cyc = 0.0
                for ins in bodies[cands[0]]:
                    mn, _, ops = ins.partition(" ")
                    if not mn.startswith("v_") or mn.startswith(("v_accvgpr", "v_nop")):
                        continue
                    c = cost(mn, ops)
                    cyc += c
                    counts["mfma" if c == 16.0 and mn.startswith("v_mfma") else "fast_2.3" if c == FAST else "slow_4.2" if c == SLOW else "other"] += 1
                n = sum(counts.values())
                res[k] = {"symbol": cands[0], "static_valu_instructions": n, "classes": counts, "mean_issue_cycles_per_valu": round(cyc / max(n, 1), 3)}
    return res


if __name__ == "__main__":
    r = analyse()
    if "--json" in sys.argv:
        print(json.dumps(r, indent=1))
    else:
        for k, v in r.items():
            print(f"{k:18s} static VALU {v['static_valu_instructions']:5d}  {v['classes']}  mean issue cycles {v['mean_issue_cycles_per_valu']}")
