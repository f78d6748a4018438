#!/usr/bin/env python3
"""Static census of one trace_kernel instance: its vector instructions by issue-cost class and by the part of
the source they come from.

    make -C ray_tracing_amd/csrc asm EXTRA=-gline-tables-only      # rt_kernels.s with .loc lines
    python tools/census.py "<0, 0, 0>" [--blocks]

Every instruction is attributed to the source line of the innermost inlined function it was generated from
(the .loc directive in force) and that line to a phase by the function it lies in (PHASES below: first and last
source line are found by searching rt_kernels.hip for the markers); an instruction from one of the one-line vector
helpers (fma_, dot, madd ...) goes to the phase of the code around it.  Issue costs are those measured by
tools/valu_rates.hip on this chip: 2 cycles for all-VGPR fp32 fma / mul / add / sub, bit operations and moves;
4 for compares, selects on a fresh mask, conversions, integer multiplies, min / max, shifts left, SDWA / DPP and
anything with a scalar-register or literal-free scalar operand; 8 for the transcendentals.  The counts are
static (one per instruction in the loop body, whatever its execution frequency): they say where the code is,
the profile's SQ_INSTS_VALU says how often it ran.
"""
import argparse
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "ray_tracing_amd", "csrc", "rt_kernels.s")
SRC = os.path.join(ROOT, "ray_tracing_amd", "csrc", "rt_kernels.hip")

# phase -> (start marker, end marker) in rt_kernels.hip; a line belongs to the last phase whose range holds it
PHASES = [
    ("vector helpers", "RT_DEV V3 v3(", "RT_DEV float rsqrt_nr"),
    ("normalise (rsqrt_nr)", "RT_DEV float rsqrt_nr", "RT_DEV float rcp_nr"),
    ("reciprocal (rcp_nr)", "RT_DEV float rcp_nr", "RT_DEV float mul_legacy_after_trans"),
    ("sqrt_rn", "RT_DEV float mul_legacy_after_trans", "RT_DEV V3 cross"),
    ("RNG", "RT_DEV uint32_t hash32", "RT_DEV float glsl_random"),
    ("RNG (shader's random)", "RT_DEV float glsl_random", "RT_DEV void sincos2pi(float u, float& c_out, float& s_out)\n{"),
    ("sincos", "RT_DEV void sincos2pi(float u, float& c_out, float& s_out)\n{", "RT_DEV V3 unit_vector"),
    ("samplers", "RT_DEV V3 unit_vector", "template <int SEM> struct Window"),
    ("closest-hit merge", "struct Best", "RT_DEV void test_sphere"),
    ("sphere test", "RT_DEV void test_sphere", "RT_DEV float hit_box"),
    ("box test", "RT_DEV float hit_box", "RT_DEV float dot_unfused"),
    ("cube test", "RT_DEV float dot_unfused", "RT_DEV bool triangle_candidate"),
    ("triangle test", "RT_DEV bool triangle_candidate", "RT_DEV bool bound_hit"),
    ("bounds", "RT_DEV bool bound_hit", "RT_DEV uint32_t quantise"),
    ("quantise", "RT_DEV uint32_t quantise", "void trace_kernel("),
    ("kernel: set-up", "void trace_kernel(", "    auto camera_origin = "),
    ("camera ray of a new sample", "    auto camera_origin = ", "    V3 gs = v3(0, 0, 0);"),
    ("kernel: set-up", "    V3 gs = v3(0, 0, 0);", "    while (alive) {"),
    ("kernel: sphere loop", "    while (alive) {", "        // The lanes still tracing, and this lane's place among them"),
    ("kernel: boxes / cubes / tree walk", "        // The lanes still tracing, and this lane's place among them", "        // ---- the surface"),
    ("kernel: surface (record, normal, material)", "        // ---- the surface", "        // Two draws per lane: the bounce's pair"),
    ("kernel: accumulate + draws", "        // Two draws per lane: the bounce's pair", "        // randomUnitVector(): the direct sphere point, or"),
    ("kernel: restart / scatter", "        // randomUnitVector(): the direct sphere point, or", "        // normalize(v), or v as it is"),
    ("kernel: new direction", "        // normalize(v), or v as it is", "    if (!TRACE && valid) {\n        if (CAN_SLICE"),
    ("kernel: write-back", "    if (!TRACE && valid) {\n        if (CAN_SLICE", "// Turns the per-tile costs"),
]

FOUR = re.compile(r"^v_(cmp|cmpx|cndmask|cvt|mul_lo|mul_hi|min|max|med3|lshl|ldexp|floor|fract|rndne|bfe|bfi|alignbit|mad_|mul_u32|mul_i32|"
                  r"readlane|readfirstlane|writelane|mbcnt|add_co|sub_co|addc|subb|lshl_add|and_or|xad|perm|lshl_or|add3|or3|div_)")
EIGHT = re.compile(r"^v_(sqrt|rsq|rcp|sin|cos|log|exp)_")


def cost_class(mn, ops):
    if EIGHT.match(mn):
        return "8"
    if FOUR.match(mn) or "sdwa" in mn or "dpp" in mn or "_sdwa" in ops or "dst_sel" in ops or "row_" in ops:
        return "4"
    if re.search(r"(^|[ ,\[])s\d+|s\[\d+:\d+\]|vcc|exec", ops.split(";")[0]) and not mn.startswith("v_cndmask"):
        return "4"            # a scalar-register operand halves the rate
    return "2"


def phase_table(src):
    text = open(src).read()
    table = []
    for name, a, b in PHASES:
        ia, ib = text.find(a), text.find(b)
        if ia < 0 or ib < 0:
            continue
        table.append((text.count("\n", 0, ia) + 1, text.count("\n", 0, ib) + 1, name))
    return table


def phase_of(line, table):
    best = "other"
    for lo, hi, name in table:
        if lo <= line < hi:
            best = name
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("instance", help='template arguments as rocprof prints them, e.g. "<0, 0, 0>"')
    ap.add_argument("--blocks", action="store_true", help="also list every basic block of the main loop")
    args = ap.parse_args()
    sem, geo, mode = [int(x) for x in re.findall(r"\d+", args.instance)]
    label = f"_ZN3rtd12trace_kernelILi{sem}ELi{geo}ELi{mode}EEE"
    table = phase_table(SRC)
    lines = open(ASM).read().split("\n")
    try:
        start = next(i for i, l in enumerate(lines) if l.startswith(label) and l.rstrip().endswith(":") is False and ":" in l)
    except StopIteration:
        sys.exit(f"{label}: not in {ASM}")
    cur_line, in_loop, depth_seen = 0, False, False
    per_phase = collections.defaultdict(lambda: collections.Counter())
    blocks, block = [], None
    have_loc = False
    sticky = "other"
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"\.loc\s+\d+\s+(\d+)", s)
        if m:
            cur_line = int(m.group(1)); have_loc = True
            continue
        if s.startswith(".LBB"):
            in_loop = "Loop" in l or "Depth" in l
            block = {"name": s.split(":")[0], "in_loop": in_loop, "counts": collections.Counter(), "phases": collections.Counter()}
            blocks.append(block)
            continue
        if not s or s.startswith((";", ".", "//")):
            if "=>" in s and block is not None:
                block["in_loop"] = True
            continue
        parts = s.split(None, 1)
        mn, ops = parts[0], parts[1] if len(parts) > 1 else ""
        if mn.startswith("v_"):
            cls = "valu" + cost_class(mn, ops)
        elif mn.startswith("s_"):
            cls = "nop" if mn == "s_nop" else ("waitcnt" if mn == "s_waitcnt" else "salu")
        elif mn.startswith("ds_"):
            cls = "lds"
        elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cls = "vmem"
        else:
            cls = "other"
        if block is not None and block["in_loop"]:
            ph = phase_of(cur_line, table) if have_loc else "(no .loc: build the asm with EXTRA=-gline-tables-only)"
            if ph == "vector helpers":
                ph = sticky
            else:
                sticky = ph
            per_phase[ph][cls] += 1
            block["counts"][cls] += 1
            block["phases"][ph] += 1
    cols = ["valu2", "valu4", "valu8", "salu", "lds", "vmem", "nop", "waitcnt"]
    print(f"trace_kernel<{sem}, {geo}, {mode}>: instructions inside the bounce loop (static), by source phase")
    print(f"{'phase':46s}" + "".join(f"{c:>9s}" for c in cols) + f"{'valu':>9s}{'cycles':>9s}")
    tot = collections.Counter()
    for ph, c in sorted(per_phase.items(), key=lambda kv: -(kv[1]['valu2'] + kv[1]['valu4'] + kv[1]['valu8'])):
        valu = c["valu2"] + c["valu4"] + c["valu8"]
        cyc = 2 * c["valu2"] + 4 * c["valu4"] + 8 * c["valu8"]
        print(f"{ph:46s}" + "".join(f"{c[k]:9d}" for k in cols) + f"{valu:9d}{cyc:9d}")
        tot.update(c)
    valu = tot["valu2"] + tot["valu4"] + tot["valu8"]
    print(f"{'total':46s}" + "".join(f"{tot[k]:9d}" for k in cols) + f"{valu:9d}{2 * tot['valu2'] + 4 * tot['valu4'] + 8 * tot['valu8']:9d}")
    if args.blocks:
        print("\nbasic blocks of the loop:")
        for b in blocks:
            if b["in_loop"] and sum(b["counts"].values()):
                c = b["counts"]
                main_phase = b["phases"].most_common(1)[0][0] if b["phases"] else ""
                print(f"  {b['name']:12s} valu {c['valu2']:3d}/{c['valu4']:3d}/{c['valu8']:2d}  salu {c['salu']:3d}  lds {c['lds']:2d}  {main_phase}")


if __name__ == "__main__":
    main()
