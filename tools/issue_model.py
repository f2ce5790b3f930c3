"""
what the search kernel's instruction stream costs in issue cycles, from its disassembly and the measured per-class
issue costs (tools/issue_rate.hip -> profiles/r3_issue_rate.json).  run in the authoring container (hipcc
cross-compiles, no GPU needed):

    python tools/issue_model.py            -> profiles/r3_issue_model.json

method: compile nimrud_amd/csrc/nm_features.hip with the Makefile's flags and --save-temps, take the dominant
instance k_scale_features<7, 3, false, true>, split it into basic blocks, price every vector instruction with the
measured cost of its class (column "4 waves per SIMD": one-wave workgroups, like the kernel's), and identify the four
blocks every wave executes once per scale by what they hold:
    phase A prologue   (the table reads and squared differences: >= 12 ds_read_b64 - pairs count twice - and >= 20 fp64
                        instructions)
    phase A            (the inclusion tests: >= 60 v_alignbit_b32)
    row walk           (>= 30 ds_read_b64 and >= 30 ds_read_b32)
    epilogue           (the eigen-solve: the largest block with >= 4 v_rcp_f64 / v_rsq_f64 and >= 40 fp64 instructions
                        that holds no library division - those are the optional covariance / normal outputs -
                        plus the moment conversion block in front of it)
the rest of a wave's instructions (cells, boxes, staging loops, branches - executed a data-dependent number of
times) is the PMC count (profiles/r3_instruction_mix.json, SQ_INSTS_VALU per wave and scale) minus the instructions
of the identified blocks, priced at the average cost of the remaining blocks' static mix.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_Z16k_scale_featuresILi7ELi3ELb0ELb1EEv"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         "-mllvm", "-disable-machine-licm", "-fconstexpr-steps=20000000", "-mllvm",
         "-amdgpu-sched-strategy=max-memory-clause"]
FOUR = ("v_alignbit", "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_lo", "v_mul_hi", "v_mul_i32_i24", "v_mul_u32_u24",
        "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev", "v_lshl_add", "v_add3", "v_bfe", "v_lshrrev_b64",
        "v_lshlrev_b64", "v_cvt", "v_cmp", "v_readlane", "v_writelane", "v_readfirstlane", "_dpp", "v_mad_u64",
        "v_lshl_or", "v_and_or", "v_ldexp", "v_div_", "v_lshl_add_u64", "v_mbcnt", "v_perm", "v_med3", "v_min_i32",
        "v_max_i32", "v_min_u32", "v_max_u32", "v_sub_co", "v_add_co", "v_addc", "v_subb", "v_bfi", "v_bfrev",
        "v_bcnt", "v_min_f32", "v_max_f32", "v_cndmask")


def load_costs():
    data = json.load(open(os.path.join(REPO, "profiles", "r3_issue_rate.json")))
    col = data["waves_per_simd"].index(4)
    c = {k: float(v[col]) for k, v in data["cycles"].items()}
    return {"f64": c["v_add_f64"], "trans": c["v_rcp_f64"], "four": c["v_alignbit_b32 (imm)"],
            "two": c["v_add_u32"]}


def price(op, cost):
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return cost["trans"], "trans"
    if "_f64" in op:
        return cost["f64"], "f64"
    if any(t in op for t in FOUR):
        return cost["four"], "four"
    return cost["two"], "two"


def main():
    cost = load_costs()
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["--save-temps", "-c",
                        os.path.join(REPO, "nimrud_amd", "csrc", "nm_features.hip"), "-o", "nm_features.o"],
                       cwd=tmp, check=True, stderr=subprocess.DEVNULL)
        text = open(os.path.join(tmp, "nm_features-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    start = text.index("\n" + KERNEL) + 1
    end = text.index(".Lfunc_end", start)
    blocks, cur = [], ["entry", []]
    for line in text[start:end].split("\n"):
        m = re.match(r"^(\.LBB[0-9_]+):", line)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
        elif line.startswith("\t") and not line.startswith(("\t;", "\t.")):
            cur[1].append(line.split()[0])
    blocks.append(cur)
    table = []
    for name, ops in blocks:
        valu = [o for o in ops if o.startswith("v_")]
        cls = {"f64": 0, "trans": 0, "four": 0, "two": 0}
        cycles = 0.0
        for o in valu:
            c, k = price(o, cost)
            cycles += c
            cls[k] += 1
        table.append({"block": name, "valu": len(valu), "classes": cls, "issue_cycles": cycles,
                      "library_division": sum(o.startswith("v_div_") for o in ops),
                      "ds_read_b64": sum(o == "ds_read_b64" for o in ops) +
                                     2 * sum(o.startswith(("ds_read2_b64", "ds_read2st64_b64")) for o in ops),
                      "ds_read_b32": sum(o == "ds_read_b32" for o in ops),
                      "alignbit": sum(o.startswith("v_alignbit") for o in ops),
                      "vmem": sum(o.startswith(("global_", "buffer_")) for o in ops)})
    role = {}
    for b in table:
        if b["alignbit"] >= 60:
            role[b["block"]] = "phase A (inclusion tests)"
        elif b["ds_read_b64"] >= 30 and b["ds_read_b32"] >= 30:
            role[b["block"]] = "row walk"
        elif b["ds_read_b64"] >= 12 and b["classes"]["f64"] >= 20:
            role[b["block"]] = "phase A prologue (centre table)"
    # (the optional covariance / normal outputs have solves of their own, with library divisions and square roots:
    # v_div_scale / v_div_fmas; the feature epilogue uses the raw v_rcp / v_rsq seeds)
    solve = [b for b in table if b["classes"]["trans"] >= 4 and b["classes"]["f64"] >= 40 and
             b["library_division"] == 0 and b["block"] not in role]
    if solve:
        main_solve = max(solve, key=lambda b: b["valu"])
        role[main_solve["block"]] = "epilogue (moments -> features, eigen-solve)"
        i = [b["block"] for b in table].index(main_solve["block"])
        if i > 0 and table[i - 1]["classes"]["f64"] >= 20 and table[i - 1]["block"] not in role:
            role[table[i - 1]["block"]] = "epilogue (conversion of the moments)"
    named = [b for b in table if b["block"] in role]
    rest = [b for b in table if b["block"] not in role and b["valu"] > 0]
    named_valu = sum(b["valu"] for b in named)
    named_cycles = sum(b["issue_cycles"] for b in named)
    rest_avg = sum(b["issue_cycles"] for b in rest) / max(sum(b["valu"] for b in rest), 1)
    mix = json.load(open(os.path.join(REPO, "profiles", "r3_instruction_mix.json")))
    valu_dyn = float(mix["per_wave"]["SQ_INSTS_VALU"][0])
    other = max(valu_dyn - named_valu, 0.0)
    out = {
        "what": __doc__.strip().split("\n\n")[0],
        "kernel": "k_scale_features<7, 3, false, true>",
        "issue_cost_per_class": cost,
        "blocks_every_wave_runs_once_per_scale": [
            {"role": role[b["block"]], **b} for b in named],
        "their_valu_instructions": named_valu,
        "their_issue_cycles": named_cycles,
        "valu_instructions_per_wave_and_scale_pmc": valu_dyn,
        "other_instructions": other,
        "other_instructions_avg_cost": rest_avg,
        "main_path_issue_cycles": named_cycles + other * rest_avg,
        "static_blocks": table,
    }
    path = os.path.join(REPO, "profiles", "r3_issue_model.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)
    for b in out["blocks_every_wave_runs_once_per_scale"]:
        print("  %-48s %4d VALU  %7.0f cycles" % (b["role"], b["valu"], b["issue_cycles"]))
    print("  other: %.0f instructions at %.2f = %.0f cycles; main path %.0f issue cycles per wave and scale"
          % (other, rest_avg, other * rest_avg, out["main_path_issue_cycles"]))


if __name__ == "__main__":
    main()
