"""Issue cost of the vector instructions of each kernel, from its ISA and the measured cost of every instruction class.

    python scripts/isa_mix.py [tag]      -> profiles/<tag>_isa_mix.json     (build container: hipcc -S per source file, no GPU needed)

Why: on gfx950 a wave64 vector instruction does NOT always hold its SIMD for 4 cycles.  scripts/probes/valu_peak.hip (profiles/r04_valu_peak.json)
measures 2.2 cycles for the plain 32-bit VOP1/VOP2 integer operations (v_add_u32, v_sub_u32, v_and / or / xor / not_b32, v_lshrrev_b32, v_ashrrev_i32,
v_mov_b32, v_bitop3_b32, v_cndmask_b32 on vcc) and 4.1-4.3 for everything else the kernels use (packed 16-bit, three-operand, DPP / SDWA, compares,
min / max, left shifts, 64-bit operations, lane reads; v_swap_b32 8).  SQ_ACTIVE_INST_VALU counts one per instruction whatever its class (the same probe
under rocprofv3, scripts/probes/valu_peak_pmc.sh), so "4 x instructions / SIMD-cycles" overstates the busy share of a kernel by up to 1.8 x: that is how
round 3's tables came to show 1.07-1.33 for some kernels.  The busy share needs the kernel's instruction MIX; the counters have no per-class split, so
the mix is taken from the kernel's text: every vector instruction weighted by 8^(loop depth) (LLVM annotates the depth of each block), classified by
the probe's table.  scripts/pmc_summarize.py multiplies SQ_INSTS_VALU by the resulting cycles per instruction.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from otter_amd import build as B  # noqa: E402

FILES = ["wfa_affine_reg.hip", "wfa_affine.hip", "myers_edit.hip", "wfa_edit.hip", "wfa_adaptive.hip", "poa.hip", "cluster.hip", "pipeline.hip"]


def cost_table():
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_valu_peak.json")))
    best = {}
    for c in d["classes"]:
        name = c["class"]
        if not name.startswith("v_") or "+" in name:
            continue
        op = name.split(" ")[0]
        key = op + (" dpp" if " dpp" in name else "") + (" sgpr" if "SGPR" in name else "")
        cyc = c["simd_cycles_per_wave_inst"]
        if key not in best or c["waves_per_simd"] > best[key][1]:
            best[key] = (cyc, c["waves_per_simd"])
    t = {k: v[0] for k, v in best.items()}
    t["v_cndmask_b32"] = 2.25          # on vcc behind the compare that wrote it: (v_cmp + v_cndmask pair: 6.4 cycles) - v_cmp 4.2; a lone chain of them reads 22.9
    return t


def classify(op, table, default=4.15):
    """op: mnemonic as printed by LLVM (with _e32 / _e64 / _dpp / _sdwa suffix)"""
    base = re.sub(r"_(e32|e64)$", "", op)
    if base.endswith("_sdwa"):
        return table.get("v_add_u32_sdwa", default)
    if base.endswith("_dpp"):
        return table.get("v_mov_b32 dpp", default)
    if base == "v_cndmask_b32" and op.endswith("_e64"):
        return table.get("v_cndmask_b32 sgpr", default)
    if base in table:
        return table[base]
    if base.startswith("v_cmp"):
        return table.get("v_cmp_gt_i32", default)
    if base.startswith("v_pk_"):
        return table.get("v_pk_add_u16", default)
    return default


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+(<[0-9, a-z]+>)?)", n)
    return m.group(1) if m else n[:40]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    table = cost_table()
    out = {}
    tmp = tempfile.mkdtemp(prefix="isa_mix_")
    procs = []
    for f in FILES:
        asm = os.path.join(tmp, f.replace(".hip", ".s"))
        procs.append((f, asm, subprocess.Popen([B.hipcc()] + [x for x in B.FLAGS if x != "-fPIC"] + ["-S", "--cuda-device-only", "-o", asm, os.path.join(B.CSRC, f)],
                                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    for f, asm, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc -S failed for " + f)
        cur, depth = None, 0
        acc = None
        for ln in open(asm, errors="replace"):
            m = re.match(r"\s*\.type\s+(\S+),@function", ln)
            if m:
                cur = m.group(1); depth = 0
                acc = {"n": 0, "w": 0.0, "wc": 0.0, "cyc": 0.0, "fast": 0, "ops": collections.Counter()}
                continue
            if cur is None:
                continue
            if re.match(r"\s*s_endpgm", ln) or ln.startswith(".Lfunc_end"):
                if acc and acc["n"]:
                    name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip() or cur
                    k = short(name)
                    out[k] = {"file": f, "valu_insts_static": acc["n"], "fast_share_static": round(acc["fast"] / acc["n"], 4),
                              "cycles_per_inst_static": round(acc["cyc"] / acc["n"], 4), "cycles_per_inst": round(acc["wc"] / acc["w"], 4),
                              "top_ops": dict(acc["ops"].most_common(12))}
                cur = None; acc = None
                continue
            m = re.match(r"\.LBB\d+_\d+:(.*)", ln)
            if m:
                d = re.search(r"Depth=(\d+)", m.group(1))
                depth = int(d.group(1)) if d else 0
                continue
            m = re.match(r"\s+(v_[a-z0-9_]+)", ln)
            if m and acc is not None:
                op = m.group(1)
                c = classify(op, table)
                w = 8.0 ** depth
                acc["n"] += 1; acc["cyc"] += c; acc["w"] += w; acc["wc"] += w * c; acc["fast"] += c < 3.0
                acc["ops"][op] += 1
    res = {"what": "estimated SIMD cycles per wave64 vector instruction of each kernel: instruction classes from profiles/r04_valu_peak.json, weights 8^(loop depth) over the kernel's ISA",
           "cost_table": {k: round(v, 3) for k, v in sorted(table.items())}, "default_cycles": 4.15, "kernels": out}
    path = os.path.join(ROOT, "profiles", "%s_isa_mix.json" % tag)
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: kv[0]):
        print("%-52s %6d insts  fast %.2f  static %.2f  loop-weighted %.2f cycles/inst" % (k[:52], v["valu_insts_static"], v["fast_share_static"], v["cycles_per_inst_static"], v["cycles_per_inst"]))
    print(path)


if __name__ == "__main__":
    main()
