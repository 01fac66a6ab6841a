#!/usr/bin/env python3
"""Instruction mix of a kernel's row loop, from hipcc's assembly (`make -C grayskull_amd/csrc asm`
writes /tmp/gs_asm/*.s).  Finds the kernel, takes its largest backward-branch loop (the unrolled
row loop), counts VALU / LDS / VMEM / SALU instructions and splits the VALU ones into the two
issue classes measured by scripts/ubench_valu.cpp (profiles/r02*_ubench_valu.log):
  full -- v_add_u32 v_sub_u32 v_subrev_u32 v_and_b32 v_or_b32 v_xor_b32 v_add_u16 v_sub_u16
          v_max_u16 v_min_u16 v_add_f32 v_mov_b32 (plain VOP1/VOP2 encodings of these)
  half -- everything else (packed 16-bit, v_perm_b32, v_alignbit_b32, multiplies, shifts, every
          VOP3 / DPP / SDWA form)
usage: isa_count.py <file.s> <kernel-name-substring> <rows per loop trip> [out.json]
"""
import collections
import json
import re
import sys

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u16", "v_sub_u16",
        "v_subrev_u16", "v_max_u16", "v_min_u16", "v_add_f32", "v_mov_b32", "v_not_b32"}


def kernel_body(lines, name):
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[_A-Za-z0-9]+:", l) and name in l and not l.startswith(".L"):
            start = i
        elif start is not None and l.strip().startswith("s_endpgm"):
            return lines[start:i + 1]
    raise SystemExit("kernel %r not found" % name)


def main():
    path, name, rows = sys.argv[1], sys.argv[2], int(sys.argv[3])
    body = kernel_body(open(path).read().splitlines(), name)
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB[0-9_]+)", l) or re.search(r"s_branch\s+(\.LBB[0-9_]+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            if best is None or i - labels[m.group(1)] > best[1] - best[0]:
                best = (labels[m.group(1)], i)
    if best is None:
        raise SystemExit("no loop found")
    mix = collections.Counter()
    cls = collections.Counter()
    for l in body[best[0]:best[1] + 1]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        mix[op] += 1
        if op.startswith("v_"):
            base = op[:-4] if op.endswith("_e32") else op  # _e64 = VOP3 encoding: half rate
            plain = base in FULL and "dpp" not in t and "sdwa" not in t
            cls["full" if plain else "half"] += 1
        elif op.startswith("ds_"):
            cls["ds_add" if op.startswith("ds_add") else "ds_other"] += 1
        elif op.startswith("buffer_") or op.startswith("global_"):
            cls["vmem"] += 1
        elif op.startswith("s_"):
            cls["salu/other"] += 1
    per_row = {k: round(v / rows, 2) for k, v in cls.items()}
    out = {"kernel": name, "loop_lines": [best[0], best[1]], "rows_per_trip": rows, "per_wave_row": per_row,
           "per_trip": dict(cls), "mnemonics_per_trip": dict(mix.most_common()),
           "source": "scripts/isa_count.py over hipcc's gfx950 assembly of grayskull_amd/csrc/gs_fused.cpp"}
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt + "\n")
    print(json.dumps({"per_wave_row": per_row, "valu_per_row": round((cls["full"] + cls["half"]) / rows, 1)}))
    for op, n in mix.most_common(24):
        print("  %-22s %5d  %.1f/row" % (op, n, n / rows))


if __name__ == "__main__":
    main()
