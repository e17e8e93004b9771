#!/usr/bin/env python3
"""Static VALU instruction mix of the two sweep loops + their modelled issue time.

Compiles ps_capi.hip to gfx950 assembly, finds the hot loop of ps_hamming_nn<2> and of every
ps_ransac_score<MODE>, counts VALU instructions per unit of work (descriptor pair / (hypothesis, match)
evaluation per wave) and prices them with the per-instruction issue costs measured on the MI355X box by
profiles/microbench/valu_rates (cycles per wave64 instruction per SIMD, normalised to 2.4 GHz).
Writes profiles/isa_mix.json; bench.py's VALU_PER_UNIT table (instructions per unit of the hot loops) is taken from it.
"""
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATES = os.path.join(ROOT, "profiles", "microbench", "valu_rates_mi355x.txt")


def load_costs():
    cost = {}
    for line in open(RATES):
        parts = line.split()
        if len(parts) >= 4 and parts[0].startswith("v_"):
            try:
                cost[" ".join(parts[:-3])] = float(parts[-1])
            except ValueError:
                pass
    return cost


def price(mn, cost):
    table = [("v_pk_", cost.get("v_pk_mul_f32", 4.4)), ("v_fma_f32", cost.get("v_fma_f32 (3 regs)", 3.0)),
             ("v_fmac_f32", cost.get("v_fmac_f32", 2.8)), ("v_mul_f32", cost.get("v_mul_f32", 2.6)),
             ("v_add_f32", cost.get("v_add_f32", 2.6)), ("v_sub_f32", cost.get("v_add_f32", 2.6)),
             ("v_rcp_f32", cost.get("v_rcp_f32", 8.3)), ("v_sqrt_f32", cost.get("v_sqrt_f32", 8.3)),
             ("v_xor_b32", cost.get("v_xor_b32", 2.4)), ("v_and_b32", cost.get("v_and_b32", 2.4)),
             ("v_add_u32", cost.get("v_add_u32", 2.4)), ("v_addc", cost.get("v_add_u32", 2.4)),
             ("v_mov_b32", cost.get("v_mov_b32", 2.3)), ("v_bcnt", cost.get("v_bcnt_u32_b32", 4.4)),
             ("v_min_u32", cost.get("v_min_u32", 4.2)), ("v_lshl_or", cost.get("v_lshl_or_b32", 4.45))]
    for pre, c in table:
        if mn.startswith(pre):
            return c
    return 4.2  # compares, min3/max3, f64, cvt, div_*, cndmask, ...: the half-rate class


def hot_loop(body, key):
    """The innermost loop block with the most occurrences of `key`."""
    blocks = re.split(r"\n(\.LBB\d+_\d+):", body)
    best = None
    for i in range(1, len(blocks), 2):
        n = blocks[i + 1].count(key)
        if best is None or n > best[0]:
            best = (n, blocks[i + 1])
    return best[1]


def valu_of(txt):
    ins = [l.split()[0] for l in txt.split("\n") if l.strip().startswith("v_")]
    return Counter(ins)


def main():
    cost = load_costs()
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
                               "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", "-mllvm", "-disable-vector-combine", "-S", "--cuda-device-only", "-o", asm, os.path.join(ROOT, "putslam_amd", "csrc", "ps_capi.hip")],
                              stderr=subprocess.DEVNULL)
        s = open(asm).read()
    out = {"cost_source": "profiles/microbench/valu_rates_mi355x.txt (cycles per wave64 instruction per SIMD @ 2.4 GHz)"}
    # Hamming sweep: the 4-query unrolled loop evaluates 4 queries x TPL(2) train rows per lane
    a = s.index("\n_ZN5psdev13ps_hamming_nnILi2EE")
    body = s[a:s.index(".Lfunc_end", a)]
    c = valu_of(hot_loop(body, "v_bcnt"))
    pairs = c["v_bcnt_u32_b32"] / 8.0
    out["ps_hamming_nn"] = {"unit": "descriptor pair per wave", "valu_per_unit": sum(c.values()) / pairs,
                            "model_cycles_per_unit": sum(price(k, cost) * v for k, v in c.items()) / pairs,
                            "mix": dict(c)}
    for mode in (0, 1, 2, 4):
        a = s.index("\n_ZN5psdev15ps_ransac_scoreILi%dEE" % mode)
        body = s[a:s.index(".Lfunc_end", a)]
        # basic blocks (explicit labels and the assembler's "; %bb.N:" fall-through blocks), grouped by the loop
        # header LLVM annotates them with ("in Loop: Header=BBx_y")
        blocks = re.split(r"\n(\.LBB\d+_\d+:|; %bb\.\d+:)", body)
        loops = {}
        names = [blocks[i] for i in range(1, len(blocks), 2)]
        texts = [blocks[i + 1] for i in range(1, len(blocks), 2)]
        for i, t in enumerate(texts):
            first = t.split("\n", 1)[0]
            if "This Inner Loop Header" in first and "Depth=1" in first:
                hdr = re.sub(r"^\.L", "", names[i]).rstrip(":")
                last = i
                for j in range(i + 1, len(texts)):
                    f2 = texts[j].split("\n", 1)[0]
                    if ("Header=" + hdr + " ") in f2 + " ":
                        last = j
                    elif "Loop Header" in f2:
                        break
                loops[hdr] = texts[i:last + 1]   # contiguous body: header .. last block annotated with this header
        reproj = mode in (1, 2)
        best = None
        for hdr, ts in loops.items():
            alltxt = "\n".join(ts)
            if reproj:
                if "v_pk_fma_f32" not in alltxt or "v_max3_f32" in alltxt:
                    continue  # want the match loop whose upper window bound is hoisted (the common one)
            else:
                if "s_load_dwordx4" not in alltxt or "v_pk_mul_f32" not in alltxt:
                    continue
            n = sum(valu_of(t).total() for t in ts)
            if best is None or n > best[0]:
                best = (n, ts)
        ts = best[1]
        tot, cold = Counter(), Counter()
        for t in ts:
            if "v_div_scale_f32" in t or "v_cvt_f64_f32" in t:
                cold += valu_of(t)   # '/' fallback and double fallback: taken only outside the window / inside the band
            else:
                tot += valu_of(t)
        per = 1.0
        if not reproj:
            per = max(1, "\n".join(ts).count("s_load_dwordx4") // 2)  # unrolled: 2 record loads per match
        out["ps_ransac_score<%d>" % mode] = {"unit": "(hypothesis, match) evaluation per wave",
                                             "valu_per_unit": sum(tot.values()) / per,
                                             "model_cycles_per_unit": sum(price(k, cost) * v for k, v in tot.items()) / per,
                                             "mix": dict(tot), "cold_fallback_valu": sum(cold.values()) / per}
    # ---- round 2 kernels ----
    # decision-exact scoring kernel: the hot path of one evaluation runs from the loop header to the add-with-carry
    a = s.index("\n_ZN5psdev20ps_ransac_score_fastILi1ELb1EE")
    body = s[a:s.index(".Lfunc_end", a)]
    end = body.rindex("v_addc_co_u32")
    start = body.rindex("This Loop Header", 0, end)
    c = valu_of(body[start:body.index("\n", end)])
    out["ps_ransac_score_fast<1>"] = {"unit": "(hypothesis, match) evaluation per wave", "valu_per_unit": sum(c.values()),
                                      "model_cycles_per_unit": sum(price(k, cost) * v for k, v in c.items()), "mix": dict(c)}
    # matrix-core Hamming sweep: the main loop = the innermost loop with 16 MFMAs (4 train tiles x 4 k-steps) per trip
    a = s.index("\n_ZN5psdev15ps_hamming_mfmaILi4EE")
    body = s[a:s.index(".Lfunc_end", a)]
    hdrs = [m.start() for m in re.finditer(r"=>This Inner Loop Header", body)]
    best = None
    for h0 in hdrs:
        seg = body[h0:]
        m = re.search(r"s_cbranch_\w+ \.LBB\d+_\d+\n(?=\.LBB|; %bb)", seg)
        nxt = body.find("Loop Header", h0 + 40)
        seg = body[h0:nxt if nxt > 0 else len(body)]
        n = seg.count("v_mfma_f32_32x32x64_f8f6f4")
        if n and (best is None or n > best[0]):
            best = (n, seg)
    c = valu_of(best[1])
    nm = c.pop("v_mfma_f32_32x32x64_f8f6f4")
    out["ps_hamming_mfma<4>"] = {"unit": "query tile x 4 train tiles per wave (4096 distances)", "mfma_per_unit": nm,
                                 "valu_per_unit": sum(c.values()), "valu_per_mfma": sum(c.values()) / nm,
                                 "note": "the block includes the cold partial-last-tile masking (16 compares/selects)",
                                 "mix": dict(c)}
    with open(os.path.join(ROOT, "profiles", "isa_mix.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        if isinstance(v, dict):
            print(k, round(v["valu_per_unit"], 1), "VALU,", round(v.get("model_cycles_per_unit", float("nan")), 1), "cycles per", v["unit"])


if __name__ == "__main__":
    main()
