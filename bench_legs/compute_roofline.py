"""Rooflines for the legs that HBM does not bound.  A kernel's label ("vector ALU", "LDS tables") is backed here by a
number: its instruction stream against the rate the chip issues that stream at.

Chip: 256 CUs x 4 SIMDs at 2.4 GHz (MI355X_MICROARCH.md).  Two ceilings:

* vector ALU -- wave64 vector instructions per second.  `per_element` = vector instructions per element and lane, counted by
  the SQ (SQ_INSTS_VALU x 64 / elements: tools/sq_counters.sh -> profiles/sq_counters.json, r6_sq_counters.txt).
  `issue_cycles` = cycles a SIMD needs per wave instruction of THIS kernel's opcode mix when nothing else holds it up: the
  static mix of the kernel's ISA (tools/isa_mix.py -> profiles/isa_mix.json, r6_isa_mix.txt) weighted with the per-class
  issue intervals measured by tools/oprate.hip (profiles/r2_oprate.txt: a plain VOP2 every 2.3-2.8 cycles with >= 4 waves per
  SIMD, VOP3 encodings / shifts / bitop3 / SDWA 4.3-4.7, v_mad_u64_u32 5.2-5.5).  peak = 1024 SIMDs x 2.4e9 / issue_cycles.
* LDS -- 64-lane LDS instructions per second.  `per_element` = LDS instructions per element and lane (SQ_INSTS_LDS); a CU's
  LDS pipe takes `cycles_per_access` per wave instruction (LDS_CYCLES below: the pipe's 256 bytes per clock for 16-byte
  reads, tools/ldsbank.hip's best patterns for the rest, profiles/r6_ldsbank.txt).  peak = 256 CUs x 2.4e9 /
  cycles_per_access.

Both JSON files are measurements committed under profiles/ (the counters on an MI355X box, the mix from the built library);
the constants below stand in for a kernel the files do not hold.  `gap` says what stands between a kernel and its ceiling
when the fraction is below 0.6."""
import json
import os

from .common import ROOT

SIMDS, CUS, CLOCK_HZ = 1024, 256, 2.4e9
# cycles of a CU's LDS pipe per 64-lane instruction.  ds_read_b128: 4.0 = 1 KiB at the pipe's 256 bytes per clock -- the C4
# reconstruct kernel itself sustains one every 4.78 cycles (profiles/r6_sq_counters.txt), where the loop of tools/ldsbank.hip
# reaches 5.2 (its xors share the issue slots); ds_write_b128 13.5 and ds_read_b32 2.3 are that probe's best patterns
LDS_CYCLES = {"ds_read_b128": 4.0, "ds_write_b128": 13.5, "ds_read_b32": 2.3}


def _load(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            return json.load(fh)["kernels"]
    except (OSError, ValueError, KeyError):
        return {}


SQ = _load("sq_counters.json")      # per kernel: valu_per_element, lds_per_element, valu_issue_cycles_measured, ...
MIX = _load("isa_mix.json")         # per kernel: mix, issue_cycles_4_waves, lds_static

# stand-ins (rounds 3-5's counters: profiles/r5_ew_sq.txt, r3_c4_sq.txt) for a kernel the two files do not hold
FALLBACK = {
    "m61_inv": {"valu": 113.0, "lds": 0.0, "issue": 3.57}, "m127_inv": {"valu": 508.0, "lds": 0.0, "issue": 3.46},
    "mont128_inv": {"valu": 626.5, "lds": 0.0, "issue": 3.80}, "gf2_128_inv": {"valu": 1882.0, "lds": 185.7, "issue": 3.48},
    "gf2_128_mul": {"valu": 411.0, "lds": 52.0, "issue": 3.45}, "c4_share": {"valu": 8027.0, "lds": 56.0, "issue": 3.72},
    "c4_recover": {"valu": 4757.0, "lds": 1280.0, "issue": 3.91}, "prg_blocks": {"valu": 254.0, "lds": 152.0, "issue": 4.41},
}
GAPS = {
    "m127_inv": "10^7 elements at 64 per lane are 2 442 waves -- 2.4 per SIMD, one round: too few to cover the memory round trips "
                "of the chain and the dependent products of the one Fermat inversion (the chain length that is fastest all the "
                "same: fewer inversions, profiles/r5_probe_inv_chain.txt; two levels since round 6: 1.7 x instead of 2.5 x the "
                "algorithmic bytes, profiles/r6_probe_inv_two_level.txt)",
    "mont128_inv": "as Mersenne127 (2.4 waves per SIMD at 10^7), the one Fermat inversion per chain is 174 dependent Montgomery products, "
                   "and since the product was halved (round 6) the rolled chain's 2.6 x traffic shows: 84 B x 41 G/s = 3.4 TB/s",
    "gf2_128_inv": "shares the SIMD with 186 LDS table accesses per element (the window tables of its products)",
    "gf2_128_mul": "shares the SIMD with 52 LDS accesses per product: 44 ds_read_b128 + 8 ds_write_b128 = 0.6 of the LDS pipe",
    "c4_share": "two waves per SIMD (the tile's coefficient registers): little latency cover for its 4.5-cycle shifts",
}


def _figures(key):
    fb = FALLBACK[key]
    sq, mix = SQ.get(key), MIX.get(key)
    return {"valu": sq["valu_per_element"] if sq else fb["valu"], "lds": sq["lds_per_element"] if sq else fb["lds"],
            "issue": mix["issue_cycles_4_waves"] if mix else fb["issue"],
            "source": ("profiles/sq_counters.json" if sq else "profiles/r5_ew_sq.txt / r3_c4_sq.txt (stand-in constants)") + " + "
                      + ("profiles/isa_mix.json" if mix else "a stand-in opcode mix") + " x profiles/r2_oprate.txt",
            "mix": mix["mix"] if mix else None, "measured_in_profile": sq}


def valu_roofline(key, elements_per_s):
    """key: "<field>_inv" | "gf2_128_mul" | "c4_share" ..; elements_per_s: what the bench just measured"""
    c = _figures(key)
    wave_instr_per_s = elements_per_s * c["valu"] / 64.0
    peak = SIMDS * CLOCK_HZ / c["issue"]
    out = {"bound": "vector ALU", "unit": "wave64 vector instr/s", "per_element": c["valu"], "achieved": wave_instr_per_s, "peak": peak,
           "frac": wave_instr_per_s / peak, "issue_cycles_of_the_mix": c["issue"], "opcode_mix": c["mix"],
           "issue_cycles_measured": SIMDS * CLOCK_HZ / wave_instr_per_s,
           "frac_of_plain_vop2_rate": wave_instr_per_s / (SIMDS * CLOCK_HZ / 2.3), "source": c["source"]}
    if key in GAPS:
        out["gap"] = GAPS[key]
    return out


def lds_roofline(key, elements_per_s):
    """the LDS pipe of a CU against the table accesses of `key` (c4_recover, prg_blocks, gf2_128_mul)"""
    c = _figures(key)
    if key == "prg_blocks":
        width, cyc = "ds_read_b32", LDS_CYCLES["ds_read_b32"]
    elif key in ("gf2_128_mul", "gf2_128_inv"):      # 44 reads + 8 writes of 16 bytes per product
        width, cyc = "ds_read_b128 / ds_write_b128 (44 : 8)", (44 * LDS_CYCLES["ds_read_b128"] + 8 * LDS_CYCLES["ds_write_b128"]) / 52
    else:
        width, cyc = "ds_read_b128", LDS_CYCLES["ds_read_b128"]
    acc_per_s = elements_per_s * c["lds"] / 64.0
    peak = CUS * CLOCK_HZ / cyc
    out = {"bound": "LDS table reads", "unit": f"64-lane {width} instr/s", "per_element": c["lds"], "achieved": acc_per_s, "peak": peak,
           "frac": acc_per_s / peak, "cycles_per_access": cyc, "source": c["source"].split(" + ")[0] + " + profiles/r6_ldsbank.txt"}
    return out
