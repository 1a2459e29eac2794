"""Rooflines for the legs that HBM does not bound.  A kernel's label ("vector ALU", "LDS tables") is backed here by a
number: its instruction stream against the rate the chip issues that stream at.

Chip: 256 CUs x 4 SIMDs at 2.4 GHz (MI355X_MICROARCH.md).  Two ceilings:

* vector ALU -- wave64 vector instructions per second.  `per_element` = vector instructions per element and lane, from the SQ
  counters (SQ_INSTS_VALU x 64 / elements; profiles/r6_sq_counters.txt, tools/sq_counters.sh).  `issue_cycles` = cycles a
  SIMD needs per wave instruction of THIS kernel's opcode mix at its occupancy: the static mix of the kernel's ISA
  (tools/isa_mix.py, profiles/r6_isa_mix.txt) weighted with the per-class issue intervals measured by tools/oprate.hip
  (profiles/r2_oprate.txt: a plain VOP2 every 2.3-2.8 cycles at >= 4 waves per SIMD, VOP3 / shifts / bitop3 4.3-4.7,
  v_mad_u64_u32 5.2-5.5; 4.5-5.1 for everything at 2 waves per SIMD).  peak = 1024 SIMDs x 2.4e9 / issue_cycles.
* LDS -- 64-lane 16-byte LDS instructions per second.  `per_element` = ds_read_b128 / ds_write_b128 per element and lane;
  a CU's LDS pipe takes `cycles_per_access` per wave instruction (tools/ldsbank.hip, profiles/r4_ldsbank.txt and
  r6_ldsbank.txt: 5.3 cycles per ds_read_b128 whatever the bank pattern, 13.8 per ds_write_b128).
  peak = 256 CUs x 2.4e9 / cycles_per_access.

Every entry names where its constants were measured; `gap` says what stands between the kernel and the ceiling when the
fraction is below 0.6."""
SIMDS, CUS, CLOCK_HZ = 1024, 256, 2.4e9

# vector-ALU legs: lane instructions per element (SQ counters), issue cycles of the kernel's own mix (ISA mix x oprate)
VALU = {
    "m61_inv": {"per_element": 113.0, "issue_cycles": 3.3, "waves_per_simd": 3,
                "source": "profiles/r5_ew_sq.txt (176.6 M wave instructions per 10^8 elements); mix: 5.2 products of 19-21 "
                          "instructions, v_mad_u64_u32 (5.3 cycles) : VOP2 (2.5) about 1 : 2.5"},
    "m127_inv": {"per_element": 642.0, "issue_cycles": 4.0, "waves_per_simd": 4.8,
                 "source": "profiles/r5_ew_sq.txt (100.3 M wave instructions per 10^7 elements)",
                 "gap": "4.8 waves per SIMD in one round: 40 % of the wave cycles wait (scratch round trips of the rolled chain)"},
    "mont128_inv": {"per_element": 1010.0, "issue_cycles": 4.0, "waves_per_simd": 4,
                    "source": "profiles/r6_sq_counters.txt",
                    "gap": "the one Fermat inversion per chain (174 products) runs at the lane's own latency"},
    "gf2_128_inv": {"per_element": 1882.0, "issue_cycles": 4.4, "waves_per_simd": 5,
                    "source": "profiles/r5_ew_sq.txt (294.0 M wave instructions per 10^7 elements; 185 LDS accesses beside them)"},
    "gf2_128_mul": {"per_element": 411.0, "issue_cycles": 4.4, "waves_per_simd": 5,
                    "source": "profiles/r5_ew_sq.txt (64.2 M wave instructions per 10^7 products: shifts, v_bitop3_b32 -- VOP3 classes)"},
    "c4_share": {"per_element": 8027.0, "issue_cycles": 3.6, "waves_per_simd": 2,
                 "source": "profiles/r3_c4_sq.txt (1.568 G wave instructions per 1.25e7 secrets of (40,13))"},
}
# LDS-table legs: 16-byte LDS accesses per element and lane
LDS = {
    "c4_recover": {"per_element": 1280.0, "cycles_per_access": 5.3,
                   "source": "profiles/r3_c4_sq.txt (250 M ds_read_b128 wave instructions per 1.25e7 secrets: 40 parties x 32 "
                             "nibble lookups), profiles/r4_ldsbank.txt (5.28-5.42 cycles per ds_read_b128)"},
    "prg_blocks": {"per_element": 160.0, "cycles_per_access": 2.5, "width": "ds_read_b32",
                   "source": "profiles/r3_c4_sq.txt / r2_pmc_aes.txt (159.5 M ds_read_b32 wave instructions per 2^26 blocks: 16 x 10 "
                             "table lookups), profiles/r2_ldsbank.txt (2.3-2.7 cycles per ds_read_b32)"},
    "gf2_128_mul": {"per_element": 52.0, "cycles_per_access": 6.6,
                    "source": "44 ds_read_b128 (5.3 cycles) + 8 ds_write_b128 (13.8) per product: profiles/r5_ew_sq.txt, r4_ldsbank.txt"},
}


def valu_roofline(key, elements_per_s):
    c = VALU[key]
    wave_instr_per_s = elements_per_s * c["per_element"] / 64.0
    peak = SIMDS * CLOCK_HZ / c["issue_cycles"]
    out = {"bound": "vector ALU", "unit": "wave64 vector instr/s", "per_element": c["per_element"],
           "achieved": wave_instr_per_s, "peak": peak, "frac": wave_instr_per_s / peak,
           "issue_cycles_of_the_mix": c["issue_cycles"], "issue_cycles_measured": SIMDS * CLOCK_HZ / wave_instr_per_s,
           "frac_of_plain_vop2_rate": wave_instr_per_s / (SIMDS * CLOCK_HZ / 2.3), "source": c["source"]}
    if "gap" in c:
        out["gap"] = c["gap"]
    return out


def lds_roofline(key, elements_per_s):
    c = LDS[key]
    acc_per_s = elements_per_s * c["per_element"] / 64.0
    peak = CUS * CLOCK_HZ / c["cycles_per_access"]
    out = {"bound": "LDS table reads", "unit": f"64-lane {c.get('width', 'ds_read_b128')} instr/s", "per_element": c["per_element"],
           "achieved": acc_per_s, "peak": peak, "frac": acc_per_s / peak, "cycles_per_access": c["cycles_per_access"],
           "source": c["source"]}
    if "gap" in c:
        out["gap"] = c["gap"]
    return out
