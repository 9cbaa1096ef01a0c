#!/usr/bin/env python3
"""Static instruction statistics of the gfx950 kernels (no GPU needed): hipcc --cuda-device-only -S on a .hip source, then per kernel
and per LOOP of a kernel (LLVM's own loop annotations in the assembly) the number of VALU / SALU / LDS / VMEM / SMEM instructions,
waits and branches.  The family-stage kernels are bound by vector-instruction issue (profiles/r03_experiments.md: VALU busy 85 %), so
the VALU count of a hot loop body predicts its time well enough to rank variants on the CPU before a GPU-minute is spent on them.

  tools/isa_stats.py fgumi_amd/csrc/fastpath.hip [-D...] [--kernel k_split_cols] [--loops] [--min-loop 8]

Counts are STATIC (instructions in the body, not executed instructions: a body with internal branches counts every side once).
Round 4 added two columns that turned out to matter as much as the VALU count: `slow` = vector instructions that issue at a quarter of the
rate (v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32, transcendentals, f64 conversions — per the ISA guide, not measured here —: counted in `valu` too) and `xmask` = exec-mask saves
(s_and_saveexec / s_or_saveexec / s_andn2_saveexec: one per divergent region the compiler built — nested conditionals over the lane number
cost k_emit more scalar than vector instructions, profiles/r04_experiments.md); a scalar instruction takes a SIMD's issue slot like a
vector one (tools/ubench/valu_rate.hip)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

CLASSES = ["valu", "salu", "lds", "vmem", "smem", "wait", "branch", "other"]
EXTRA = ["slow", "xmask"]          # sub-counts (of valu / salu), printed behind the classes
SLOW = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_lo_i32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_exp_", "v_sin_", "v_cos_",
        "v_cvt_f64_", "v_cvt_u32_f64", "v_cvt_i32_f64", "v_cvt_f32_f64")      # (f64 add / mul / fma issue at the full rate on this part: tools/ubench/valu_rate.hip)


def classify(op):
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_endpgm") or op.startswith("s_setpc") or op.startswith("s_swappc"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store") or op.startswith("s_memtime") or op.startswith("s_dcache"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def parse(asm):
    """-> {kernel: dict(total=Counter, loops={header: dict(depth, parent, counts=Counter, line)})}"""
    kernels = {}
    cur = None
    block_loop = None          # loop header the current basic block belongs to (innermost)
    fn_re = re.compile(r"^([A-Za-z_][\w.$]*):\s*; @")
    bb_re = re.compile(r"^(\.LBB\d+_\d+):\s*(?:;\s*(.*))?$")
    for ln, line in enumerate(asm.split("\n"), 1):
        m = fn_re.match(line)
        if m:
            cur = dict(total=collections.Counter(), loops=collections.OrderedDict())
            kernels[m.group(1)] = cur
            block_loop = None
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        m = bb_re.match(line)
        if m:
            label, note = m.group(1), m.group(2) or ""
            hdr = re.search(r"Loop Header: Depth=(\d+)", note)
            inl = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", note)
            if hdr:
                depth = int(hdr.group(1))
                cur["loops"][label] = dict(depth=depth, counts=collections.Counter(), line=ln)
                block_loop = label
            elif inl:
                block_loop = ".L" + inl.group(1)
                if block_loop not in cur["loops"]:
                    cur["loops"][block_loop] = dict(depth=int(inl.group(2)), counts=collections.Counter(), line=ln)
            else:
                block_loop = None
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        k = classify(op)
        ks = [k]
        if k == "valu" and op.startswith(SLOW):
            ks.append("slow")
        if k == "salu" and "saveexec" in op:
            ks.append("xmask")
        for kk in ks:
            cur["total"][kk] += 1
            if block_loop is not None:
                cur["loops"][block_loop]["counts"][kk] += 1
    return kernels


def main():
    args = sys.argv[1:]
    if not args:
        print(__doc__)
        return 1
    src = args[0]
    want = None
    show_loops = "--loops" in args
    min_loop = 8
    flags = []
    i = 1
    while i < len(args):
        a = args[i]
        if a == "--kernel":
            want = args[i + 1]; i += 1
        elif a == "--min-loop":
            min_loop = int(args[i + 1]); i += 1
        elif a != "--loops":
            flags.append(a)
        i += 1
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-DFGX_HAVE_CODEC",
               "--cuda-device-only", "-S", src, "-o", out, "-w"] + flags
        subprocess.check_call(cmd)
        asm = open(out).read()
    kernels = parse(asm)
    names = demangle(list(kernels))
    print(f"# {src} {' '.join(flags)}  (static counts; gfx950)")
    print(f"{'kernel':60s} " + " ".join(f"{c:>7s}" for c in CLASSES + EXTRA))
    for k, d in kernels.items():
        nm = names[k]
        nm = nm.replace("(anonymous namespace)::", "").replace("fgx::", "")
        nm = re.sub(r"^void ", "", re.sub(r"\(.*", "", nm))
        if "rocprim" in nm or "hipcub" in nm:
            continue
        if want and want not in nm:
            continue
        print(f"{nm[:60]:60s} " + " ".join(f"{d['total'][c]:7d}" for c in CLASSES + EXTRA))
        if show_loops:
            for h, L in d["loops"].items():
                n = sum(L["counts"][c] for c in CLASSES)
                if n < min_loop:
                    continue
                print(f"    loop {h:12s} depth {L['depth']}  " + " ".join(f"{c}={L['counts'][c]}" for c in CLASSES + EXTRA if L["counts"][c]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
