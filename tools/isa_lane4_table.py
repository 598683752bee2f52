"""Per-section instruction table of the SHIPPED lane decoder kernel (lz4hip::decode_lane4_kernel<true,192,32,128,2,2,2,16>).

Builds lz4hip_api.hip with -DLZ4HIP_SECTION_MARKERS (every LZ4HIP_SECTION("...") of lz4hip_decode_lane4.hpp becomes a comment line in
the listing), walks the kernel's listing in layout order and attributes every instruction to the marker before it.  The loop body exists
twice (the flushing iteration and the one that requests input), so every section appears twice; rare paths (the byte-wise parser, the end of
a block, the exact tail) are sections of their own and are left out of the per-iteration sum.  Instructions are classified as
  VOP3  vector-ALU, 64-bit encoding (v_cndmask_b32_e64, v_perm_b32, v_alignbyte_b32, v_lshl_add_u32, v_and_or_b32, v_add3_u32, v_bfe_u32, *_e64 ...)
  VOP2  vector-ALU, 32-bit encoding (v_add_u32, v_and_b32, v_mov_b32, v_cmp_*_e32 ...)
  SALU / LDS / VMEM / WAIT
usage: python tools/isa_lane4_table.py [out.txt]      (CPU only: hipcc cross-compiles)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN6lz4hip19decode_lane4_kernelILb1ELi192ELi32ELi128ELi2ELi2ELi2ELi16EEEvNS_5BatchEiPKjij"
VOP3_ONLY = ("v_perm_b32", "v_alignbyte_b32", "v_alignbit_b32", "v_lshl_add_u32", "v_add_lshl_u32", "v_and_or_b32", "v_or3_b32", "v_add3_u32", "v_bfe_u32", "v_bfe_i32",
             "v_bfi_b32", "v_lshl_or_b32", "v_mad_u32_u24", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64",
             "v_lshl_add_u64", "v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32", "v_readlane_b32", "v_writelane_b32", "v_min3_u32", "v_max3_u32", "v_med3_u32", "v_xad_u32",
             "v_cmp_class", "v_sad_u32", "v_msad_u8")


def classify(op, text):
    if op.startswith("v_"):
        if op.endswith("_e64") or op.startswith(VOP3_ONLY) or op.endswith("_sdwa") or op.endswith("_dpp"):
            return "VOP3"
        if op.startswith("v_cmp") and not op.endswith("_e32"):
            # v_cmp_xx vcc, ... is VOPC (32-bit); with an SGPR destination the listing says _e64
            return "VOP2"
        if op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co")) and "vcc" not in text:
            return "VOP3"
        return "VOP2"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "WAIT"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "VMEM"
    return "OTHER"


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    with tempfile.TemporaryDirectory() as td:
        s_file = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "lz4net_amd", "csrc"), "-DLZ4HIP_SECTION_MARKERS",
               "-S", "--cuda-device-only", os.path.join(ROOT, "lz4net_amd", "csrc", "lz4hip_api.hip"), "-o", s_file]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        lines = open(s_file, errors="replace").read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    seen = collections.Counter()
    sec = "prologue (one block per lane: set-up)"
    table = collections.OrderedDict()
    mnem = collections.defaultdict(collections.Counter)
    meta = {}
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r";\s*@@SECTION (.*)", t)
        if m:
            name = m.group(1).strip()
            seen[name] += 1
            copy = (seen[name] + (1 if name.startswith("T4 parse") else 0)) // (2 if name.startswith("T4 parse") else 1)
            sec = "%s [copy %d]" % (name, copy)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = classify(op, t)
        table.setdefault(sec, collections.Counter())[c] += 1
        mnem[sec][op] += 1
    for l in lines[end:end + 400]:
        m = re.search(r"\.(num_vgpr|numbered_sgpr|private_seg_size), (\d+)", l)
        if m and KERNEL in l:
            meta[m.group(1)] = int(m.group(2))
    rows = []
    rows.append("# lz4hip::decode_lane4_kernel<true,192,32,128,2,2,2,16> -- static instruction counts of the compiled kernel by source section")
    rows.append("# (tools/isa_lane4_table.py: hipcc -O3 -DLZ4HIP_SECTION_MARKERS; the loop body exists twice: copy 1 = the iteration that flushes, copy 2 = the one that requests input)")
    rows.append("# registers: %s" % meta)
    rows.append("%-52s %5s %5s %5s %4s %5s %5s" % ("section", "VOP3", "VOP2", "SALU", "LDS", "VMEM", "WAIT"))
    hot = [collections.Counter(), collections.Counter()]
    RARE = ("T4x", "B5 ", "prologue")
    for name, c in table.items():
        rows.append("%-52s %5d %5d %5d %4d %5d %5d" % (name, c["VOP3"], c["VOP2"], c["SALU"], c["LDS"], c["VMEM"], c["WAIT"]))
        if not name.startswith(RARE):
            k = 0 if "[copy 1]" in name else 1
            hot[k].update(c)
    for k in (0, 1):
        c = hot[k]
        rows.append("%-52s %5d %5d %5d %4d %5d %5d   <- always-executed sections, VALU %d" % ("SUM copy %d (without T4x trap, B5 end of block)" % (k + 1), c["VOP3"], c["VOP2"], c["SALU"], c["LDS"], c["VMEM"], c["WAIT"], c["VOP3"] + c["VOP2"]))
    both = hot[0] + hot[1]
    rows.append("per iteration (mean of the two copies): VALU %.0f (VOP3 %.0f, VOP2 %.0f), SALU %.0f, LDS %.1f, VMEM %.1f" % (
        (both["VOP3"] + both["VOP2"]) / 2, both["VOP3"] / 2, both["VOP2"] / 2, both["SALU"] / 2, both["LDS"] / 2, both["VMEM"] / 2))
    rows.append("")
    rows.append("# most frequent mnemonics per hot section (both copies summed)")
    agg = collections.defaultdict(collections.Counter)
    for name, c in mnem.items():
        if not name.startswith(RARE):
            agg[re.sub(r" \[copy \d\]", "", name)].update(c)
    for name, c in agg.items():
        rows.append("%-44s %s" % (name, ", ".join("%s x%d" % kv for kv in c.most_common(14))))
    text = "\n".join(rows)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
