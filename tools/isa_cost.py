#!/usr/bin/env python3
"""Price the headline kernel's instruction stream with the measured per-instruction issue costs.

  sum over VALU mnemonics of (dynamic count x measured cost at 4 waves/SIMD)  vs  the kernel's measured duration

Inputs, all committed under profiles/ (nothing here runs on a GPU):
  * the kernel's ISA: disassembled from phaneron_amd/lib/libphaneron_hip.so (llvm-objdump --offloading);
  * the dynamic totals per launch: SQ_INSTS_VALU, SQ_INSTS_LDS, SQ_INSTS_SALU ... from profiles/rNN_pmc_sq.txt;
  * the cost of each mnemonic: profiles/rNN_opbench3.jsonl rows at waves_per_simd == 4, cus == 256 (wall ns per
    wave64 instruction per SIMD with the whole chip busy, i.e. including the clock the chip sustains for that class).

The dynamic MIX is taken from the static histogram of the kernel's loop bodies (every instruction that lies inside
a backward branch's range): the loops are the per-layer / per-pixel-group bodies and run the same number of times,
the straight-line prologue runs once.  The mix is then scaled to the counter's dynamic total.

  python tools/isa_cost.py [--kernel SUBSTR] [--ms 0.0545]
"""
import argparse
import collections
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
HEADLINE = "fused_v210_combine_lds_kernelILi4ELi6ELi1024ELb0ELi0ELb0E"  # N = 4, P = 6, 1024 lanes, shipped form (no pipelining, whole kernel, lines without tails)

# mnemonics opbench3 did not time, priced as the measured instruction of the same hardware class
SAME_AS = {
    "v_fmac_f32": "v_fmac_f32", "v_fma_f32": "v_fma_f32", "v_mul_f32": "v_mul_f32", "v_add_f32": "v_add_f32",
    "v_sub_f32": "v_sub_f32", "v_subrev_f32": "v_sub_f32", "v_mov_b32": "v_mov_b32", "v_and_b32": "v_and_b32",
    "v_or_b32": "v_or_b32", "v_xor_b32": "v_and_b32", "v_lshrrev_b32": "v_lshrrev_b32", "v_add_u32": "v_add_u32",
    "v_sub_u32": "v_add_u32", "v_subrev_u32": "v_add_u32", "v_cndmask_b32": "v_cndmask_b32",
    "v_lshlrev_b32": "v_lshlrev_b32", "v_bfe_u32": "v_bfe_u32", "v_and_or_b32": "v_and_or_b32",
    "v_lshl_or_b32": "v_lshl_or_b32", "v_lshl_add_u32": "v_lshl_add_u32", "v_add_lshl_u32": "v_lshl_add_u32",
    "v_add3_u32": "v_add3_u32", "v_or3_b32": "v_and_or_b32", "v_max_f32": "v_max_f32", "v_min_f32": "v_max_f32",
    "v_med3_f32": "v_med3_f32", "v_rndne_f32": "v_rndne_f32", "v_cvt_f32_u32": "v_cvt_f32_u32",
    "v_cvt_u32_f32": "v_cvt_u32_f32", "v_cvt_f32_ubyte0": "v_cvt_f32_ubyte0", "v_mad_u32_u24": "v_mad_u32_u24",
    "v_mul_u32_u24": "v_mad_u32_u24", "v_perm_b32": "v_perm_b32", "v_min_u32": "v_max_f32", "v_max_u32": "v_max_f32",
    "v_min_i32": "v_max_f32", "v_max_i32": "v_max_f32", "v_cmp": "v_cmp_gt_f32", "v_ashrrev_i32": "v_lshrrev_b32",
    "v_bfi_b32": "v_and_or_b32", "v_readfirstlane_b32": "v_mov_b32", "v_mul_lo_u32": "v_mul_lo_u32",
    "v_mul_hi_u32": "v_mul_lo_u32", "v_lshlrev_b64": "v_lshl_add_u32", "v_lshl_add_u64": "v_lshl_add_u32",
    "v_accvgpr_write_b32": "v_mov_b32", "v_accvgpr_read_b32": "v_mov_b32", "v_pk_fma_f32": "v_pk_fma_f32",
    "v_pk_mul_f32": "v_pk_fma_f32", "v_pk_add_f32": "v_pk_fma_f32", "v_frexp_exp_i32_f32": "v_rndne_f32",
    "v_ffbh_u32": "v_rndne_f32", "v_sub_co_u32": "v_add_u32", "v_add_co_u32": "v_add_u32", "v_addc_co_u32": "v_add_u32",
    "v_subb_co_u32": "v_add_u32", "v_mad_u64_u32": "v_mul_lo_u32", "v_alignbit_b32": "v_perm_b32",
    "v_rcp_f32": "v_rcp_f32", "v_exp_f32": "v_rcp_f32", "v_log_f32": "v_rcp_f32", "v_sqrt_f32": "v_rcp_f32",
}
FALLBACK_NS = 1.78  # the slow class: what every unmeasured, unmapped mnemonic is charged


def device_asm(lib):
    """Disassembly of every gfx950 bundle inside `lib` (llvm-objdump drops the bundles next to the input: work on a copy)."""
    tmp = tempfile.mkdtemp(prefix="isa_cost_")
    copy = os.path.join(tmp, os.path.basename(lib))
    subprocess.check_call(["cp", lib, copy])
    subprocess.check_call([OBJDUMP, "--offloading", copy], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = ""
    for b in sorted(glob.glob(copy + ".*gfx950")):
        text += subprocess.check_output([OBJDUMP, "-d", b]).decode()
    subprocess.call(["rm", "-rf", tmp])
    return text


def kernel_body(asm, name):
    out, on = [], False
    for line in asm.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            on = name in m.group(2) and not m.group(2).endswith(".kd")
            continue
        if on and line.strip():
            out.append(line)
    return out


def parse(body):
    """[(address, mnemonic, operands, branch target or None)]"""
    ins = []
    for line in body:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
        if not m:
            continue
        mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        target = None
        if mn.startswith("s_cbranch") or mn == "s_branch":
            t = re.search(r"\+0x([0-9a-f]+)>", line)
            target = ("rel", int(t.group(1), 16)) if t else None
        ins.append([addr, mn, ops, target])
    base = ins[0][0]
    for i in ins:
        if i[3]:
            i[3] = base + i[3][1]
    return ins


def pmc(path):
    vals = {}
    for line in open(path):
        f = line.split()
        if len(f) >= 2 and re.match(r"^[A-Z_0-9]+$", f[0]):
            try:
                vals[f[0]] = float(f[1])
            except ValueError:
                pass
    return vals


def costs(path):
    table = {}
    for line in open(path):
        r = json.loads(line)
        if r.get("waves_per_simd") == 4 and r.get("cus") == 256:
            table[r["instr"]] = (r["ns_wall_per_instr_per_simd"], r["cyc_per_instr_per_simd"])
    return table


def price(mn, table):
    key = mn
    for suffix in ("_e32", "_e64", "_sdwa", "_dpp"):
        if key.endswith(suffix):
            key = key[:-len(suffix)]
    sdwa = mn.endswith("_sdwa")
    if key.startswith("v_cmp") or key.startswith("v_cmpx"):
        key = "v_cmp"
    ref = SAME_AS.get(key)
    if sdwa and "v_or_b32_sdwa" in table:
        return table["v_or_b32_sdwa"][0], "v_or_b32_sdwa"
    if ref and ref in table:
        return table[ref][0], ref
    if key in table:
        return table[key][0], key
    return FALLBACK_NS, "unmeasured (slow class)"


def latest(pattern):
    f = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return f[-1] if f else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=HEADLINE)
    ap.add_argument("--lib", default=os.path.join(ROOT, "phaneron_amd", "lib", "libphaneron_hip.so"))
    ap.add_argument("--ms", type=float, default=None, help="measured average launch duration (default: profiles/rNN_bench.json)")
    ap.add_argument("--pmc", default=latest("r*_pmc_sq.txt"))
    ap.add_argument("--opbench", default=latest("r*_opbench3.jsonl"))
    ap.add_argument("--simds", type=int, default=1024)
    args = ap.parse_args()

    ins = parse(kernel_body(device_asm(args.lib), args.kernel))
    loops = [(i[3], i[0]) for i in ins if i[3] is not None and i[3] <= i[0]]
    hot = [i for i in ins if any(lo <= i[0] <= hi for lo, hi in loops)]
    table = costs(args.opbench)
    counters = pmc(args.pmc)
    ms = args.ms
    if ms is None:
        b = latest("r*_bench.json")
        ms = json.loads(open(b).read().strip().splitlines()[-1])["roofline"]["avg_launch_ms"]

    valu = collections.Counter(i[1] for i in hot if i[1].startswith("v_"))
    n_valu = sum(valu.values())
    mean_ns, rows = 0.0, []
    for mn, c in valu.most_common():
        ns, why = price(mn, table)
        mean_ns += ns * c / n_valu
        rows.append({"mnemonic": mn, "static_in_loops": c, "share": round(c / n_valu, 4), "ns_per_wave_instr_per_simd": ns, "priced_as": why})
    dyn = counters.get("SQ_INSTS_VALU")
    per_simd = dyn / args.simds
    predicted_us = per_simd * mean_ns * 1e-3
    lds = counters.get("SQ_INSTS_LDS", 0.0)
    # the LDS pipe is one per CU: opbench3's "random" rows are cycles per SIMD-instruction with 4 SIMDs issuing, so a
    # CU serves one wave-read every (row / 4) cycles; the kernel's gathers are data-dependent LUT reads (random)
    lds_static = collections.Counter(i[1] for i in hot if i[1].startswith("ds_"))
    n_lds = max(sum(lds_static.values()), 1)
    lds_row = {"ds_read_b32": "ds_read_b32 random (+1 mad per read)", "ds_read_u16": "ds_read_u16 random (+1 mad per read)",
               "ds_read_b64": "ds_read_b64 random (+1 mad per read)"}
    lds_ns = sum(c / n_lds * table[lds_row.get(mn, lds_row["ds_read_b32"])][0] / 4.0 for mn, c in lds_static.items())
    fast_ns = table["v_fmac_f32"][0]
    cus = args.simds // 4
    lds_us = lds / cus * lds_ns * 1e-3
    doc = {
        "kernel": args.kernel, "instructions_static": len(ins), "instructions_in_loops": len(hot), "loops": len(loops),
        "valu_static_in_loops": n_valu,
        "other_static_in_loops": dict(collections.Counter(i[1].split("_")[0] + "_" + i[1].split("_")[1] for i in hot if not i[1].startswith("v_")).most_common(8)),
        "dynamic_per_launch": {"SQ_INSTS_VALU": dyn, "SQ_INSTS_LDS": lds, "SQ_INSTS_SALU": counters.get("SQ_INSTS_SALU"),
                               "source": os.path.basename(args.pmc)},
        "valu_per_simd": round(per_simd, 1), "mean_ns_per_valu_instr": round(mean_ns, 4),
        "valu_issue_time_us": round(predicted_us, 2), "measured_launch_us": round(ms * 1e3, 2),
        "valu_issue_over_measured": round(predicted_us / (ms * 1e3), 3),
        "valu_issue_time_if_every_instruction_were_fast_class_us": round(per_simd * fast_ns * 1e-3, 2),
        "lds_static_in_loops": dict(lds_static), "lds_reads_per_cu": round(lds / cus, 1),
        "lds_ns_per_wave_read_per_cu": round(lds_ns, 3), "lds_pipe_time_us": round(lds_us, 2),
        "lds_pipe_over_measured": round(lds_us / (ms * 1e3), 3),
        "lds_counters": {"SQ_LDS_IDX_ACTIVE_per_cu": round(counters.get("SQ_LDS_IDX_ACTIVE", 0) / cus, 0),
                         "SQ_LDS_BANK_CONFLICT_per_cu": round(counters.get("SQ_LDS_BANK_CONFLICT", 0) / cus, 0)},
        "cost_source": os.path.basename(args.opbench) + " rows waves_per_simd=4, cus=256",
    }
    print(json.dumps(doc))
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
