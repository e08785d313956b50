#!/usr/bin/env python3
"""Static per-category instruction histogram of one kernel of the gfx950 code object in mh_icp.o (llvm-objdump), so that the
instruction counters of a PMC run (SQ_INSTS_VALU and its typed sub-counters) have names: which buckets the vector
instructions of a kernel fall into, including the ones no typed counter covers (compares, selects, moves, DPP, bit ops).
Usage: tools/isa_histogram.py <kernel name substring> [<kernel name substring> ...]  -> markdown on stdout
("dump the ISA of the hot loop and commit a per-category histogram under profiles/")."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

CATS = [
    ("fp64 arithmetic", r"^v_(add|mul|fma|max|min|cvt_f64|rcp|rsq|sqrt|ldexp|frexp|trunc|floor|ceil|rndne|fract|div)_?.*f64|^v_cvt_f32_f64"),
    ("fp32 add/sub/mul/fma/min/max (incl. packed)", r"^v_(pk_)?(add|sub|subrev|mul|fma|mac|fmac|mad|max|min|max3|min3|med3)_(legacy_)?f32"),
    ("conversions / floor / rounding", r"^v_(cvt_|floor_|trunc_|ceil_|rndne_|fract_)"),
    ("fp compares", r"^v_cmpx?_\w+_f(32|64)|^v_cmp_class"),
    ("integer compares", r"^v_cmpx?_\w+_[iu](16|32|64)"),
    ("selects (v_cndmask)", r"^v_cndmask"),
    ("moves (v_mov, incl. DPP / readlane / writelane / permute)", r"^v_mov_|^v_readlane|^v_readfirstlane|^v_writelane|^v_perm|^v_swap|^v_accvgpr"),
    ("64-bit integer (add / shift / mad_u64)", r"^v_(lshl_add_u64|lshlrev_b64|lshrrev_b64|ashrrev_i64|mad_[iu]64|add_co|addc_co|subb?_co|subrev_co)"),
    ("integer multiply (mul_lo / mul_hi / mad 24)", r"^v_(mul_lo|mul_hi|mul_u32_u24|mul_i32_i24|mad_[iu]32_[iu]24|mad_u32_u16)"),
    ("32-bit integer add / sub / min / max", r"^v_(add|sub|subrev|add3|min|max|min3|max3|med3|sad)_(nc_)?[iu](16|32)|^v_add3_u32|^v_lshl_add_u32|^v_add_lshl_u32|^v_xad_u32"),
    ("bit operations (and / or / xor / shifts / bfe / ffb / bitop3 / popcount)", r"^v_(and|or|xor|not|bfe|bfi|bfm|lshl|lshr|ashr|lshlrev|lshrrev|ashrrev|ffbl|ffbh|bcnt|bitop3|or3|and_or|lshl_or|alignbit|alignbyte|mbcnt)"),
    ("LDS (ds_*)", r"^ds_"),
    ("vector memory (global / buffer / flat / scratch)", r"^(global|buffer|flat|scratch)_"),
    ("scalar ALU", r"^s_(?!waitcnt|nop|cbranch|branch|endpgm|load|buffer_load|barrier|sleep|setprio|sendmsg|getpc|setpc|swappc|code_end|dcache|icache|inst_prefetch|clause|delay)"),
    ("scalar memory (s_load / s_buffer_load)", r"^s_(load|buffer_load)"),
    ("branches", r"^s_(cbranch|branch|setpc|swappc)"),
    ("waits / nops (s_waitcnt, s_nop)", r"^s_(waitcnt|nop|sleep|barrier|delay)"),
]


def disassemble():
    obj = os.path.join(ROOT, "mola_lidar_odometry_amd", "csrc", "mh_icp.o")
    tmp = tempfile.mkdtemp(prefix="isa_")
    subprocess.check_call(["cp", obj, os.path.join(tmp, "mh_icp.o")])
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "mh_icp.o"], cwd=tmp, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
    return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, co)], text=True).splitlines()


def kernel_body(lines, needle):
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            if start is not None:
                return name, lines[start + 1:i]
            if needle in m.group(1):
                start, name = i, m.group(1)
    return (name, lines[start + 1:]) if start is not None else (None, [])


def main():
    lines = disassemble()
    for needle in sys.argv[1:]:
        name, body = kernel_body(lines, needle)
        if not name:
            print(f"(no kernel matching {needle})")
            continue
        ops = [l.split()[0] for l in body if l.strip() and not l.strip().startswith("//") and re.match(r"^\s+[a-z]", l)]
        ops = [re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", o) for o in ops]
        raw = [l.split()[0] for l in body if l.strip() and re.match(r"^\s+[a-z]", l)]
        n_dpp = sum(1 for o in raw if o.endswith("_dpp"))
        hist, examples = collections.Counter(), collections.defaultdict(collections.Counter)
        for o in ops:
            for cat, rx in CATS:
                if re.search(rx, o):
                    hist[cat] += 1
                    examples[cat][o] += 1
                    break
            else:
                hist["other"] += 1
                examples["other"][o] += 1
        total = sum(hist.values())
        valu = sum(v for k, v in hist.items() if k not in ("LDS (ds_*)", "vector memory (global / buffer / flat / scratch)", "scalar ALU",
                                                          "scalar memory (s_load / s_buffer_load)", "branches", "waits / nops (s_waitcnt, s_nop)"))
        print(f"### `{name}`\n")
        print(f"{total} instructions in the code object, {valu} of them vector ALU ({n_dpp} with a DPP modifier).  Static counts: a loop body counts once.\n")
        print("| category | instructions | share | most frequent |")
        print("|---|---|---|---|")
        for cat, v in hist.most_common():
            ex = ", ".join(f"`{o}` {c}" for o, c in examples[cat].most_common(4))
            print(f"| {cat} | {v} | {100.0 * v / total:.1f} % | {ex} |")
        print()


if __name__ == "__main__":
    main()
