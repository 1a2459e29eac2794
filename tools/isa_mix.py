#!/usr/bin/env python3
"""Static opcode mix of kernels in libscl_hip.so -> the vector-issue ceiling their label "vector ALU" is priced against.

The library's device code sits in its .hip_fatbin section as clang offload bundles (one per translation unit); this tool cuts
the gfx950 code objects out, disassembles them with llvm-objdump and, for every kernel whose (demangled) name contains one of
the needles, counts vector-ALU instructions by issue class.  The classes and their issue intervals are tools/oprate.hip's
(profiles/r2_oprate.txt, cycles per wave64 instruction and SIMD with >= 4 waves resident | with 2):

    vop2     plain two-operand integer / logic / move with register or inline-constant operands      2.5 | 4.8
    vop3     everything encoded VOP3 / with an SGPR or literal operand: shifts by a register, bfe,
             alignbit, perm, bitop3, add3, lshl_add, mul_lo, mad_u32_u24, cndmask, DPP / SDWA forms  4.5 | 4.7
    mad64    v_mad_u64_u32 / v_mad_i64_i32 (the 32 x 32 -> 64 multiply-add of the limb products)      5.3 | 5.5

`issue_cycles` of a kernel = the mix-weighted mean interval: what a SIMD needs per instruction of this stream when nothing
else (memory, LDS, dependencies) holds it up.  Static counts stand in for executed counts: the kernels are straight-line
unrolled arithmetic inside one loop, and the SQ counters (tools/sq_counters.sh) give the executed totals beside them.

    python3 tools/isa_mix.py [needle ...] > profiles/r6_isa_mix.txt      (also writes profiles/isa_mix.json)
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "secure-computation-library_amd", "scl_amd", "libscl_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# key of bench_legs/compute_roofline.py -> needle in the demangled kernel name
DEFAULT_NEEDLES = {
    "m61_inv": "k_ew_inv<sclhip::M61, false",
    "m127_inv": "k_ew_inv_blocked<sclhip::M127, sclhip::FieldArith<sclhip::M127>, false, 16, 4",   # (what 10^7 elements take: chains of 64, blocks of 4)
    "mont128_inv": "k_ew_inv_rolled<sclhip::Mont128",
    "gf2_128_inv": "k_ew_inv_rolled<sclhip::Gf128",
    "gf2_128_mul": "k_ew_gf128_mul<64, 3>",
    "c4_share": "k_share_gf_tiles<13>",
    "c4_recover": "k_recover_gf128_pos<512, 2",
    "prg_blocks": "k_prg_blocks",
}
CYCLES = {"vop2": (2.5, 4.8), "vop3": (4.5, 4.7), "mad64": (5.3, 5.5)}
VOP2_FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32",
             "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32",
             "v_min_u32", "v_max_u32", "v_ashrrev_i32", "v_not_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}


def code_objects():
    """the gfx950 ELF images inside the library's .hip_fatbin section"""
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "fatbin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", SO, raw], check=True)
        blob = open(raw, "rb").read()
    out, pos = [], 0
    while True:
        at = blob.find(MAGIC, pos)
        if at < 0:
            break
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        off = at + len(MAGIC) + 8
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24: off + 24 + tl].decode()
            off += 24 + tl
            if "gfx950" in triple and sz:
                out.append(blob[at + o: at + o + sz])
        pos = at + len(MAGIC)
    return out


def classify(mn, ops):
    if not mn.startswith("v_") or mn.startswith(("v_mfma", "v_cmp", "v_readlane", "v_readfirstlane", "v_writelane", "v_nop")):
        return "vcmp" if mn.startswith("v_cmp") else None
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mn)
    if base.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
        return "mad64"
    if mn.endswith(("_dpp", "_sdwa", "_e64_dpp")):
        return "vop3"
    if base in ("v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32") and mn.endswith("_e32"):
        return "vop2"     # VOP2 with the implicit vcc carry
    if base == "v_cndmask_b32" and mn.endswith("_e32"):
        return "vop2"     # VOP2 with the implicit vcc select (the 22-cycle line of r2_oprate.txt is that probe's own vcc dependency)
    if base in VOP2_FAST and not mn.endswith("_e64"):
        # an SGPR or a 32-bit literal as a source costs the VOP3 interval (profiles/r2_oprate.txt: "v_and_b32 sgpr" 4.35)
        srcs = ops.split(",")[1:]
        if any(re.match(r"\s*(s\d+|s\[|vcc|exec|ttmp)", x) for x in srcs):
            return "vop3"
        return "vop2"
    return "vop3"


def main():
    needles = dict(DEFAULT_NEEDLES)
    for a in sys.argv[1:]:
        needles[a] = a
    found = {}
    with tempfile.TemporaryDirectory() as d:
        for i, elf in enumerate(code_objects()):
            path = os.path.join(d, f"co{i}.elf")
            open(path, "wb").write(elf)
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", "--no-show-raw-insn", path],
                                 capture_output=True, text=True, check=True).stdout
            cur = None
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
                if m:
                    name = m.group(1)
                    cur = None
                    for key, nd in needles.items():
                        if nd in name and key not in found:
                            cur = found.setdefault(key, {"kernel": name.split("(")[0].replace("void ", ""), "classes": Counter(),
                                                          "opcodes": Counter(), "lds": Counter(), "vmem": 0, "salu": 0, "total": 0})
                    continue
                if cur is None:
                    continue
                t = ln.strip().split(None, 1)
                if not t or t[0].startswith(("//", ";")):
                    continue
                mn, ops = t[0], (t[1] if len(t) > 1 else "")
                cur["total"] += 1
                if mn.startswith("ds_"):
                    cur["lds"][mn] += 1
                elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
                    cur["vmem"] += 1
                elif mn.startswith("s_"):
                    cur["salu"] += 1
                c = classify(mn, ops)
                if c and c != "vcmp":
                    cur["classes"][c] += 1
                    cur["opcodes"][re.sub(r"_(e32|e64)$", "", mn)] += 1
                elif c == "vcmp":
                    cur["classes"]["vop3"] += 1
                    cur["opcodes"]["v_cmp*"] += 1
    report = {}
    print("# tools/isa_mix.py: static vector-ALU opcode mix of the kernels HBM does not bound, and the issue interval of that mix")
    print("# (classes and intervals: profiles/r2_oprate.txt; columns: >= 4 waves per SIMD | 2 waves per SIMD)")
    for key in needles:
        k = found.get(key)
        if not k:
            print(f"{key:14s} NOT FOUND ({needles[key]})")
            continue
        n = sum(k["classes"].values())
        mix = {c: k["classes"][c] / n for c in CYCLES}
        hi = sum(mix[c] * CYCLES[c][0] for c in CYCLES)
        lo = sum(mix[c] * CYCLES[c][1] for c in CYCLES)
        top = ", ".join(f"{o} {100 * v / n:.0f}%" for o, v in k["opcodes"].most_common(6))
        print(f"{key:14s} {k['kernel'][:70]:70s} {n:6d} vector ALU of {k['total']:6d} instr  vop2 {mix['vop2']:.2f} vop3 {mix['vop3']:.2f} "
              f"mad64 {mix['mad64']:.2f}  -> {hi:.2f} | {lo:.2f} cycles per instruction   lds {sum(k['lds'].values())} vmem {k['vmem']} salu {k['salu']}")
        print(f"{'':14s} {top}")
        report[key] = {"kernel": k["kernel"], "valu_static": n, "instructions_static": k["total"], "mix": mix,
                       "issue_cycles_4_waves": hi, "issue_cycles_2_waves": lo, "lds_static": dict(k["lds"]), "vmem_static": k["vmem"]}
    with open(os.path.join(ROOT, "profiles", "isa_mix.json"), "w") as fh:
        json.dump({"_comment": "tools/isa_mix.py over libscl_hip.so; read by bench_legs/compute_roofline.py", "kernels": report}, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
