#!/usr/bin/env python3
"""Instruction mix and resources of the hot kernels, from the compiler's own assembly (no GPU needed).

    python scripts/isa_mix.py [--build] [--asm FILE] [kernel-substring ...]

--build compiles simka_ctx.hip with -save-temps into /tmp/simka_isa and reads the gfx950 .s from there.  For every kernel whose
mangled name contains one of the substrings (default: the hot ones) it prints
  * the resource table of the code object (VGPRs, AGPRs, SGPRs, spills, scratch bytes, static LDS),
  * the instruction histogram of the WHOLE kernel by issue class, and
  * the same for its innermost loops that contain a marker instruction (the 64-bit LDS compare-and-swap of the count kernels, the
    LDS atomic adds of the pair kernels, ...), found from the backward branches of the assembly,
priced with the measured issue cost per class (profiles/r05_valu_rate.txt, scripts/ubench/valu_rate.hip): "fast" VALU ~2.5 cycles per
wave64 instruction and SIMD (add, sub, and, or, xor, not, right shifts, mov, bitop3, f32 mul / fma), every other VALU instruction ~4.3
(left shifts, alignbit, bfrev, bfe, perm, every three-operand integer op, min / max, multiplies, compares, 64-bit shifts, mad_u64_u32,
mbcnt, readlane, DPP, SDWA), SALU ~4.3 (partly hidden behind another wave's VALU)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
        "v_bitop3_b32", "v_mul_f32", "v_fma_f32", "v_add_f32", "v_sub_f32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
COST = {"valu_fast": 2.5, "valu_slow": 4.3, "valu_cndmask": 4.0, "valu_f64": 8.6, "salu": 4.3}
DEFAULT = ["k_skm_count_fastILb1", "k_skm_scanILi16ELb1ELb0", "k_skm_chunksort", "k_groupILi256ELb0", "k_pairsILb0ELi1024", "k_pairs_tm", "k_tile_major", "k_skm_count_wide_fast"]
MARKERS = ["ds_cmpst_rtn_b64", "ds_add_u64", "ds_add_u32", "ds_add_rtn_u32", "ds_cmpst_rtn_b32"]


def klass(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    if op.startswith("v_"):
        if base.startswith("v_cndmask"):
            return "valu_cndmask"
        if base.endswith("_f64") or "f64" in base:
            return "valu_f64"
        if op.endswith("_dpp") or op.endswith("_sdwa"):
            return "valu_slow"
        return "valu_fast" if base in FAST else "valu_slow"
    if op.startswith("s_"):
        if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_endpgm"):
            return "sync"
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            return "branch"
        if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or op.startswith("s_memrealtime"):
            return "smem"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def parse(asm_path):
    """-> {kernel: [(label or None, op, text), ...]}, {kernel: resources}"""
    kernels, cur, res = {}, None, {}
    entries = []
    for line in open(asm_path, errors="replace"):
        if line.startswith("amdhsa.kernels:") or line.startswith("\t.amdgpu_metadata") or line.startswith("\t.section\t.rodata") or line.startswith("\t.data"):
            cur = None
        m = re.match(r"^(_Z\w+|k_\w+):\s*(;.*)?$", line)
        if m and "@function" not in line and ("; @" in line):
            cur = m.group(1); kernels[cur] = []; continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                cur = None; continue
            m = re.match(r"^(\.LBB\d+_\d+):", line)
            if m:
                kernels[cur].append((m.group(1), None, "")); continue
            m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*)$", line)
            if m and not line.lstrip().startswith(".") and not line.lstrip().startswith(";"):
                kernels[cur].append((None, m.group(1), m.group(2).split(";")[0].strip()))
            continue
        # code-object metadata: one "  - .agpr_count: ..." entry per kernel, fields in alphabetical order (.name in the middle)
        if re.match(r"^  - \.agpr_count:", line):
            entry = {}; entries.append(entry)
        m = re.match(r"^\s+(?:- )?\.(agpr_count|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|max_flat_workgroup_size):\s+(\d+)", line)
        if m and entries:
            entries[-1][m.group(1)] = int(m.group(2))
        m = re.match(r"^    \.name:\s+(\S+)", line)
        if m and entries:
            entries[-1]["name"] = m.group(1)
    for e in entries:
        if "name" in e:
            res[e["name"]] = e
    return kernels, res


def histogram(insts):
    c = collections.Counter(); ops = collections.Counter()
    for lab, op, _ in insts:
        if op:
            c[klass(op)] += 1; ops[op] += 1
    return c, ops


def loops(insts):
    """innermost loops: (start index, end index) for every backward branch whose body holds no other backward branch"""
    pos = {lab: i for i, (lab, op, _) in enumerate(insts) if lab}
    spans = []
    for i, (lab, op, txt) in enumerate(insts):
        if op and (op.startswith("s_cbranch") or op.startswith("s_branch")):
            t = txt.strip()
            if t in pos and pos[t] < i:
                spans.append((pos[t], i))
    inner = [s for s in spans if not any(o != s and s[0] <= o[0] and o[1] <= s[1] for o in spans)]
    return inner, spans


def cycles(c):
    return sum(COST.get(k, 0) * v for k, v in c.items())


def show(name, c, ops, top=14):
    valu = c["valu_fast"] + c["valu_slow"] + c["valu_cndmask"] + c["valu_f64"]
    print("    %-34s VALU %4d (fast %d, slow %d, cndmask %d, f64 %d)  SALU %4d  LDS %3d  VMEM %3d  branch %3d  sync %3d   ~%.0f issue cycles (VALU %.0f)" % (
        name, valu, c["valu_fast"], c["valu_slow"], c["valu_cndmask"], c["valu_f64"], c["salu"], c["lds"], c["vmem"], c["branch"], c["sync"], cycles(c),
        cycles({k: v for k, v in c.items() if k.startswith("valu")})))
    print("      " + ", ".join("%s %d" % (o, n) for o, n in ops.most_common(top)))


def main():
    args = sys.argv[1:]
    asm = None
    if "--build" in args:
        args.remove("--build")
        d = "/tmp/simka_isa"; os.makedirs(d, exist_ok=True)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unused-result", "-save-temps=obj",
                        "-o", d + "/lib.so", "simka_ctx.hip", "-lz", "-ldl"], cwd=os.path.join(ROOT, "simka_amd", "csrc"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = d + "/simka_ctx-hip-amdgcn-amd-amdhsa-gfx950.s"
    if "--asm" in args:
        i = args.index("--asm"); asm = args[i + 1]; del args[i:i + 2]
    if asm is None:
        sys.exit("need --build or --asm FILE")
    want = args or DEFAULT
    kernels, res = parse(asm)
    print("# instruction mix of the hot kernels (hipcc -O3, gfx950), priced with scripts/ubench/valu_rate.hip: " + ", ".join("%s %.1f" % kv for kv in COST.items()) + " cycles per wave64 instruction and SIMD")
    print("%-64s %5s %5s %6s %7s %7s %8s %9s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "scratch", "LDS(st.)", "threads"))
    sel = [k for k in kernels if any(w in k for w in want)]
    for k in sel:
        r = res.get(k, {})
        print("%-64s %5s %5s %6s %7s %7s %8s %9s" % (k[:64], r.get("vgpr_count", "?"), r.get("sgpr_count", "?"), r.get("vgpr_spill_count", "?"), r.get("sgpr_spill_count", "?"),
                                                     r.get("private_segment_fixed_size", "?"), r.get("group_segment_fixed_size", "?"), r.get("max_flat_workgroup_size", "?")))
    for k in sel:
        insts = kernels[k]
        c, ops = histogram(insts)
        print("\n%s" % k)
        show("whole kernel (static)", c, ops)
        inner, _ = loops(insts)
        for a, b in inner:
            body = insts[a:b + 1]
            bops = [op for _, op, _ in body if op]
            marks = [m for m in MARKERS if m in bops]
            if not marks or len(bops) < 12:
                continue
            c2, ops2 = histogram(body)
            show("loop %s..+%d [%s x%d]" % (insts[a][0], len(bops), marks[0], bops.count(marks[0])), c2, ops2)


if __name__ == "__main__":
    main()
