#!/usr/bin/env python3
"""Static vector-instruction mix of the hot loops and of the per-hypothesis prologue (round 4 rewrite).

Compiles ps_capi.hip (and profiles/microbench/prologue_isa.hip) to gfx950 assembly with the library's flags and counts, per
basic block, the VALU instructions of
  * the evaluation blocks of every scoring kernel in the default build (ps_ransac_score_fast<1 / 2>: one block = one
    (hypothesis, match) evaluation per lane, ending in the add-with-carry that counts; its pre-test block = TWO matches in one
    direction; ps_ransac_score_euclid<0 / 4>: one block = two packed steps = four matches; the value-exact ps_ransac_score<M>),
  * the main loop of the matrix-core matcher (per query tile and wave: 16 MFMAs) and of its VALU twin,
  * ps_stage_reorder's vote loop, and
  * the prologue's parts (sampler, Umeyama + SVD with its sweep loop, general inverse): static counts and the number of
    division / square-root expansions by kind.  The DYNAMIC count per wavefront is measured, not modelled: SQ_INSTS_VALU of
    stage 0's models-only launch (one work-group per pair) in profiles/<tag>/sq_counters_by_grid.json.
bench.py reads VALU_PER_UNIT / PK_PER_UNIT from the JSON this writes (profiles/isa_mix.json)."""
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-fno-slp-vectorize", "-mllvm", "-disable-vector-combine", "-S", "--cuda-device-only"]


def asm_of(src, td, name):
    out = os.path.join(td, name)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, src], stderr=subprocess.DEVNULL)
    return open(out).read()


def body_of(s, sym):
    a = s.index("\n" + sym)
    return s[a:s.index(".Lfunc_end", a)]


def blocks_of(body):
    parts = re.split(r"\n(\.LBB\d+_\d+:|; %bb\.\d+:)", body)
    return [(parts[i], parts[i + 1]) for i in range(1, len(parts), 2)]


def valu(txt):
    return Counter(re.sub(r"_e(32|64)$", "", l.split()[0]) for l in txt.split("\n") if l.strip().startswith("v_"))


def summary(c):
    n = sum(c.values())
    pk = sum(v for k, v in c.items() if k.startswith("v_pk_"))
    return {"valu": n, "packed": pk, "mix": dict(c)}


def main():
    out = {"source": "hipcc -S of putslam_amd/csrc/ps_capi.hip and profiles/microbench/prologue_isa.hip with the library's flags"}
    with tempfile.TemporaryDirectory() as td:
        s = asm_of(os.path.join(ROOT, "putslam_amd", "csrc", "ps_capi.hip"), td, "capi.s")
        pro = asm_of(os.path.join(ROOT, "profiles", "microbench", "prologue_isa.hip"), td, "pro.s")
    # ---- decision-exact reprojection kernels: evaluation block (ends in v_addc) and pre-test block (packed, no v_addc) ----
    for mode in (1, 2):
        for kind, label in ((0, "stage0"), (1, "stage1")):
            bl = blocks_of(body_of(s, "_ZN5psdev20ps_ransac_score_fastILi%dELb1ELi%dEE" % (mode, kind)))
            evals = [valu(t) for _, t in bl if "v_addc_co_u32" in t and t.count("v_pk_") >= 8]
            pre = [valu(t) for _, t in bl if "v_addc_co_u32" not in t and t.count("v_pk_") >= 12 and "ds_" not in t]
            e = {"unit": "(hypothesis, match) evaluation per wave"}
            if evals:
                # a block holds one evaluation per add-with-carry (the hot loop takes a whole PAIR of matches per trip)
                per_block = max(1, evals[0]["v_addc_co_u32"])
                sm = summary(evals[0])
                e.update({"valu": sm["valu"] / per_block, "packed": sm["packed"] / per_block,
                          "mix": {k: v / per_block for k, v in sm["mix"].items()}})
                e["evaluations_per_block"] = per_block
                e["evaluation_blocks"] = len(evals)
            if pre and kind == 1:
                e["pretest"] = dict(summary(pre[0]), unit="TWO matches, one direction, per wave")
            out["ps_ransac_score_fast<%d> %s" % (mode, label)] = e
    # ---- decision-exact Euclidean kernels: the block with the clamped indicator FMAs ----
    for mode in (0, 4):
        bl = blocks_of(body_of(s, "_ZN5psdev22ps_ransac_score_euclidILi%dELi1EE" % mode))
        hot = [(t.count("clamp"), valu(t)) for _, t in bl if "clamp" in t]
        hot.sort(key=lambda x: -x[0])
        steps = hot[0][0]   # packed steps in the block (one clamp each), two matches per step
        out["ps_ransac_score_euclid<%d>" % mode] = dict(summary(hot[0][1]), unit="block of %d packed steps = %d matches per wave" % (steps, 2 * steps),
                                                        valu_per_match=sum(hot[0][1].values()) / (2.0 * steps),
                                                        packed_per_match=sum(v for k, v in hot[0][1].items() if k.startswith("v_pk_")) / (2.0 * steps))
    # ---- value-exact kernels: the innermost match loop (hot path = its blocks without the '/' and double fallbacks) ----
    for mode in (0, 1, 2, 4):
        bl = blocks_of(body_of(s, "_ZN5psdev15ps_ransac_scoreILi%dEE" % mode))
        loops = {}
        for n, t in bl:
            first = t.split("\n", 1)[0]
            m = re.search(r"Header=(BB\d+_\d+)", first)
            hdr = m.group(1) if m else (n.strip(".:L") if "Loop Header" in first else None)
            if hdr:
                loops.setdefault(hdr, []).append(t)
        best = max(loops.values(), key=lambda ts: sum(t.count("s_load_dwordx4") for t in ts))   # the loop that streams match records
        hot = Counter()
        for t in best:
            if "v_div_scale_f32" not in t and "v_cvt_f64_f32" not in t:
                hot += valu(t)
        per = max(1, "\n".join(best).count("s_load_dwordx4") // 2) if mode in (0, 4) else 1
        out["ps_ransac_score<%d>" % mode] = dict(summary(hot), unit="match loop body per wave (%d matches)" % per, valu_per_match=sum(hot.values()) / per)
    # ---- matchers ----
    bl = blocks_of(body_of(s, "_ZN5psdev21ps_hamming_mfma_fusedILi4EE"))
    mf = max((t for _, t in bl), key=lambda t: t.count("v_mfma"))
    c = valu(mf)
    nm = sum(v for k, v in c.items() if k.startswith("v_mfma"))
    for k in [k for k in c if k.startswith("v_mfma")]:
        del c[k]
    out["ps_hamming_mfma_fused<4>"] = dict(summary(c), unit="query tile x 4 train tiles per wave (4096 distances)", mfma_per_unit=nm)
    bl = blocks_of(body_of(s, "_ZN5psdev13ps_hamming_nnILi2EE"))
    hb = max((t for _, t in bl), key=lambda t: t.count("v_bcnt"))
    c = valu(hb)
    out["ps_hamming_nn"] = dict(summary(c), unit="block of %d descriptor pairs per wave" % (c["v_bcnt_u32_b32"] // 8),
                                valu_per_pair=sum(c.values()) / (c["v_bcnt_u32_b32"] / 8.0))
    # ---- prologue parts ----
    for sym in ("k_sample", "k_umeyama", "k_inverse"):
        bl = blocks_of(body_of(pro, sym + ":"))
        tot, loop = Counter(), Counter()
        for n, t in bl:
            c = valu(t)
            tot += c
            if "in Loop:" in t.split("\n", 1)[0] or "Loop Header" in t.split("\n", 1)[0]:
                loop += c
        full_div = tot["v_div_fixup_f32"]
        out["prologue " + sym] = {"static_valu": sum(tot.values()), "static_valu_in_loops": sum(loop.values()),
                                  "full_ieee_divisions": full_div, "reciprocal_chains": tot["v_rcp_f32"],
                                  "square_roots": tot["v_sqrt_f32"], "moves_and_selects": tot["v_mov_b32"] + tot["v_cndmask_b32"],
                                  "note": "static: both sides of every window check are in the count (short form AND the '/' fallback); "
                                          "the dynamic figure per wavefront is measured (sq_counters_by_grid.json, stage 0's models-only launch)"}
    with open(os.path.join(ROOT, "profiles", "isa_mix.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        if isinstance(v, dict):
            print(k, {a: b for a, b in v.items() if a not in ("mix", "note", "pretest")}, ("pretest %s" % {a: b for a, b in v["pretest"].items() if a != "mix"}) if "pretest" in v else "")


if __name__ == "__main__":
    main()
