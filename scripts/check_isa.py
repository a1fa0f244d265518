#!/usr/bin/env python3
"""ISA gate for the row kernels.

Takes the SHIPPED library (wholegraph_amd/libwholegraph.so), extracts its gfx950 code object, disassembles it with
llvm-objdump and checks, per kernel family, the shape of the code hipcc produced:

  * loads in flight: the kernel must contain a straight-line stretch (no label, no branch) in which at least
    `min_loads` wide loads (global_load_dwordx4, or for converting kernels any global_load) are issued with no
    `s_waitcnt vmcnt` between them — the batch of a tile really is in flight together;
  * VALU budget: the static number of VALU instructions of the whole kernel (an upper bound for any tile of a kernel whose
    hot path has no loop; for looping kernels: of the hottest loop body) stays under `max_valu`;
  * v_mov share: register shuffling (what a conditionally defined load result costs) stays under `max_mov_share`;
  * scratch: kernels with `max_scratch` set must not spill more than that many bytes per lane (code-object metadata).

Exit status 0 = every rule holds; the table is printed either way.  Usage: check_isa.py [path/to/libwholegraph.so]
Needs only the ROCm LLVM tools (clang-offload-bundler, llvm-objdump): runs on a box without a GPU.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("WM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

# Known exception, stated rather than hidden: the converting kernel for int64 -> int16 / int8 element pairs (an int64 table
# read as a narrower integer, or the mirror scatter). Only the low dword of each loaded qword is used, hipcc recycles the
# unused HIGH register of the first load as a temporary and must wait for that load first: 1 + 3 loads in flight instead
# of 4. No shipped configuration takes this pair.
EXCEPTIONS = [(r"rows_convert_kernel<(long, (short|signed char)|(short|signed char), long), ", dict(min_loads=3))]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel-name regex (demangled) -> rule
RULES = [
    # the single-batch kernel: 4 x 1 KiB loads back to back, < 120 VALU per tile (static count of the whole kernel)
    (r"rows_batch_kernel<", dict(min_loads=4, wide=True, max_valu=120, max_mov_share=0.45, scope="kernel")),
    (r"rows_copy16_fast_kernel<", dict(min_loads=4, wide=True, max_valu=120, max_mov_share=0.45, scope="block")),
    (r"rows_flat_kernel<", dict(min_loads=4, wide=True, max_valu=160, max_mov_share=0.45, scope="block")),
    (r"rows_copy_kernel<", dict(min_loads=4, wide=False, max_valu=160, max_mov_share=0.45, scope="block")),
    (r"rows_convert_kernel<", dict(min_loads=4, wide=False, max_valu=400, max_mov_share=0.45, scope="block")),
    (r"rows_staged_gather_kernel<", dict(min_loads=4, wide=True, max_valu=400, max_mov_share=0.45, scope="block")),
    (r"rows_staged_scatter_kernel<", dict(min_loads=4, wide=True, max_valu=400, max_mov_share=0.45, scope="block")),
    # gradient apply: a batch of the tile kernel = 2 x kU row loads (gradient + table [+ states]) back to back; no scratch
    (r"step_tile_kernel<[^>]*, 16, false>", dict(min_loads=4, wide=True, max_valu=600, max_mov_share=0.45, scope="block", max_scratch=0)),
    # ... its RAGGED instantiation (round 6: dim % 4 != 0 — 513, 129, 127 floats: the same batch on the first dim / 4 pieces)
    (r"step_tile_kernel<[^>]*, 16, true>", dict(min_loads=4, wide=True, max_valu=700, max_mov_share=0.45, scope="block", max_scratch=0)),
    # ... and its 8-byte-piece instantiation (round 6: fp32 rows of whole 8-byte pieces, 602 floats): the same batch of dwordx2
    (r"step_tile_kernel<[^>]*, 8, false>", dict(min_loads=4, wide=False, load_re=r"^global_load_dwordx2\b", max_valu=600, max_mov_share=0.45,
                                         scope="block", max_scratch=0)),
    # the tree fold of long runs: 4 gradient rows per thread in flight (round 3 shipped one)
    (r"tree_fold_kernel<", dict(min_loads=4, wide=True, max_valu=600, max_mov_share=0.45, scope="block", max_scratch=0)),
    # round 5, the split sort of the owner-side ids (split_sort.cuh). What these kernels need from the compiler:
    #  * the ids of a tile / the keys of a bucket are loaded as one batch (>= 4 loads with no wait between them);
    #  * the bucket kernel and the small-tile scatter kernel stay within 64 VGPRs — two 1024-thread workgroups per CU is what
    #    hides their barriers — the bucket kernel without scratch (the scatter kernel is allowed the 56 bytes hipcc spills at
    #    that budget: measured equal to the 80-VGPR build at one workgroup per CU less);
    #  * the histogram kernel: no scratch, loads batched.
    (r"split::split_hist_kernel<", dict(min_loads=4, wide=False, max_valu=4000, max_mov_share=1.0, scope="block", max_scratch=0)),
    (r"split::split_scatter_kernel<\w[\w ]*, 12, 3, false>", dict(min_loads=4, wide=False, max_valu=4000, max_mov_share=1.0, scope="block",
                                                        max_scratch=64, max_vgprs=64)),
    (r"split::split_scatter_kernel<\w[\w ]*, 24, [35], (false|true)>", dict(min_loads=4, wide=False, max_valu=4000, max_mov_share=1.0, scope="block",
                                                        max_scratch=16, max_vgprs=128)),
    (r"split::split_sort_kernel<", dict(min_loads=4, wide=False, max_valu=4000, max_mov_share=1.0, scope="block", max_scratch=0,
                                        max_vgprs=64)),
]


def extract_code_object(so_path, workdir):
    """the gfx950 code object embedded in a host shared library (.hip_fatbin section -> offload bundle -> ELF)"""
    fat = os.path.join(workdir, "fatbin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so_path, fat])
    data = open(fat, "rb").read()
    out = []
    # a library linked from several objects carries several bundles back to back (each starts with the magic)
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    for k, st in enumerate(starts):
        piece = os.path.join(workdir, "bundle%d" % k)
        end = starts[k + 1] if k + 1 < len(starts) else len(data)
        open(piece, "wb").write(data[st:end])
        co = os.path.join(workdir, "co%d.o" % k)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co, "--allow-missing-bundles"],
                           capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def disassemble(co):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", co],
                                  text=True)
    return txt


def kernel_metadata(co):
    """{mangled kernel name: (vgprs, spilled vgprs, scratch bytes per lane)} from the code object's AMDGPU metadata note"""
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out = {}
    for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        num = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
        out[name.group(1)] = (num("vgpr_count"), num("vgpr_spill_count"), num("private_segment_fixed_size"))
    return out


def demangle(names):
    if not names:
        return {}
    import shutil
    tool = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
    r = subprocess.run([tool], input="\n".join(names) + "\n", capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


FUNC_RE = re.compile(r"^(?:[0-9a-f]+ )?<([^>]+)>:\s*$")
LABEL_RE = re.compile(r"^<[^>]+>:\s*$|^\s*<?L\w+>?:\s*$")


def split_functions(txt):
    funcs, cur, name = {}, None, None
    for line in txt.splitlines():
        m = FUNC_RE.match(line.strip())
        if m and not m.group(1).startswith("L"):
            name = m.group(1)
            cur = funcs.setdefault(name, [])
            continue
        if cur is None:
            continue
        ln = line.strip()
        if not ln or ln.startswith("//") or ln.startswith(";"):
            continue
        cur.append(ln)
    return funcs


def analyse(lines, wide, load_re=None):
    """returns (best loads-in-flight stretch, its VALU count, static VALU of the kernel, v_mov count, block VALU)"""
    load_pat = re.compile(load_re if load_re else (r"^global_load_dwordx4\b" if wide else r"^global_load_"))
    blocks, cur = [], []
    for ln in lines:
        op = ln.split()[0]
        if ln.startswith("<") or ln.endswith(">:"):   # a local label starts a new block
            if cur:
                blocks.append(cur)
            cur = []
            continue
        cur.append(ln)
        if op.startswith("s_cbranch") or op.startswith("s_branch") or op == "s_endpgm" or op.startswith("s_setpc"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    best, best_block = 0, None
    for b in blocks:
        run = 0
        top = 0
        for ln in b:
            op = ln.split()[0]
            if load_pat.match(ln):
                run += 1
                top = max(top, run)
            elif op == "s_waitcnt" and "vmcnt" in ln:
                run = 0
        if top > best:
            best, best_block = top, b
    valu = [ln for b in blocks for ln in b if ln.startswith("v_")]
    movs = [ln for ln in valu if ln.startswith("v_mov_") or ln.startswith("v_accvgpr")]
    block_valu = len([ln for ln in (best_block or []) if ln.startswith("v_")])
    return best, len(valu), len(movs), block_valu


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "wholegraph_amd", "libwholegraph.so")
    if not os.path.exists(so):
        print("check_isa: %s not found (build first: make -C wholegraph_amd/csrc)" % so)
        return 2
    failures, rows = [], []
    with tempfile.TemporaryDirectory() as wd:
        cos = extract_code_object(so, wd)
        if not cos:
            print("check_isa: no gfx950 code object in %s" % so)
            return 2
        seen = set()
        for co in cos:
            funcs = split_functions(disassemble(co))
            meta = kernel_metadata(co)
            names = demangle([n for n in funcs])
            for mangled, lines in funcs.items():
                dn = names.get(mangled, mangled)
                for pat, rule in RULES:
                    if not re.search(pat, dn) or "[clone" in dn:
                        continue
                    seen.add(pat)
                    rule = dict(rule)
                    for epat, over in EXCEPTIONS:
                        if re.search(epat, dn):
                            rule.update(over)
                    loads, valu, movs, block_valu = analyse(lines, rule["wide"], rule.get("load_re"))
                    counted = valu if rule["scope"] == "kernel" else block_valu
                    share = movs / max(valu, 1)
                    # (the v_mov share only means something for a kernel of some size: a 30-instruction kernel whose owner
                    # chain moves 8 kernel arguments into VGPRs is not "shuffling registers")
                    vgprs, spilled, scratch = meta.get(mangled, (-1, -1, -1))
                    ok = loads >= rule["min_loads"] and counted < rule["max_valu"] and (share <= rule["max_mov_share"] or valu < 60)
                    if "max_scratch" in rule and scratch > rule["max_scratch"]:
                        ok = False
                    if "max_vgprs" in rule and vgprs > rule["max_vgprs"]:
                        ok = False
                    rows.append((dn.replace("wm::(anonymous namespace)::", "").replace("(wm::(anonymous namespace)::rows_params)", ""),
                                 loads, counted, rule["scope"], valu, share, vgprs, scratch, ok))
                    if not ok:
                        failures.append(dn)
        for pat, _ in RULES:
            if pat not in seen:
                failures.append("no kernel matches " + pat)
    rows.sort()
    print("%-92s %5s %6s %-6s %6s %5s %5s %7s  %s" % ("kernel", "loads", "VALU", "scope", "static", "mov%", "VGPRs", "scratch", "gate"))
    for dn, loads, counted, scope, valu, share, vgprs, scratch, ok in rows:
        print("%-92s %5d %6d %-6s %6d %4.0f%% %5d %7d  %s" % (dn[:92], loads, counted, scope, valu, 100 * share, vgprs, scratch,
                                                              "ok" if ok else "FAIL"))
    if failures:
        print("\ncheck_isa: %d failure(s)" % len(failures))
        for f in failures[:20]:
            print("  " + f)
        return 1
    print("\ncheck_isa: %d kernels, all rules hold" % len(rows))
    return 0


if __name__ == "__main__":
    sys.exit(main())
