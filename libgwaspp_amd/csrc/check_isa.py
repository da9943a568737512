#!/usr/bin/env python3
"""Build-time guard for k_scan_cc_pipe (gw_kernels.hip): the continuous-stream scan keeps loads in flight
across loop iterations with inline-asm global_load_dwordx4 and hand-counted `s_waitcnt vmcnt(2*UNROLL)`.
That is only correct while the compiler

  * keeps exactly 2*UNROLL stream loads (and no other vector-memory instruction) between two consecutive
    counted waits,
  * never reads or writes a stream-buffer register between the load that fills it and the counted wait that
    covers it (no v_mov shuffling, no re-use as an address temporary, no spill), on the loop's back edge too,
  * reaches `s_waitcnt vmcnt(0)` on every way out of the loop before the buffers are touched again,
  * and uses no scratch memory in the kernel.

A compiler bump can break any of these without a CPU-side symptom, so the Makefile runs this on the built
object and fails the build otherwise.  Usage: check_isa.py build/gw_kernels.o

Second guard (round 4), `check_isa.py --no-scratch build/gw_boost_fp4.o` (and, with a pattern, the int8 objects): the
pre-filtered pair kernels of every matrix-core family and k_boost_finish must use NO scratch memory -- code-object metadata
private_segment_fixed_size == 0, no spilled VGPR / SGPR, and not one scratch_ instruction in the ISA.  Round 3's
kernels spilled 56 - 84 bytes per lane unnoticed (3.7 GB of scratch writes per config-5 launch)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
INST = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
FUNC = re.compile(r"^[0-9a-f]+ <(\S+)>:")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
BRANCH_TARGET = re.compile(r"<[^>]*\+0x([0-9a-f]+)>")


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.devnull],
                       check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET,
                        "--input=" + fat, "--output=" + co], check=True)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True,
                              text=True).stdout


def functions(text):
    out, name, start = {}, None, 0
    for ln in text.splitlines():
        m = FUNC.match(ln)
        if m:
            name = m.group(1)
            out[name] = {"base": int(ln.split()[0], 16), "insts": []}
            continue
        if name is None:
            continue
        m = INST.match(ln)
        if m:
            tgt = BRANCH_TARGET.search(ln)
            out[name]["insts"].append({"op": m.group(1), "args": m.group(2), "addr": int(m.group(3), 16),
                                       "target": (out[name]["base"] + int(tgt.group(1), 16)) if tgt else None})
    return out


def vregs(args):
    regs = set()
    for m in VREG.finditer(args):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def is_vmem(op):
    return op.startswith(("global_", "buffer_", "flat_", "scratch_"))


def check_pipe(name, insts, unroll, nt):
    """-> list of problems (empty = fine)."""
    bad = []
    n_loads = 2 * unroll
    if any(i["op"].startswith("scratch_") for i in insts):
        bad.append("scratch memory in use (spill)")
    is_stream = lambda i: i["op"] == "global_load_dwordx4" and (i["args"].rstrip().endswith(" nt") == nt)
    waits = [k for k, i in enumerate(insts) if i["op"] == "s_waitcnt" and i["args"].strip() == f"vmcnt({n_loads})"]
    if len(waits) != 2:
        return bad + [f"expected 2 counted waits vmcnt({n_loads}) (one per buffer), found {len(waits)}"]
    w1, w2 = waits
    by_addr = {i["addr"]: k for k, i in enumerate(insts)}

    def group_before(w):
        """indices of the vector-memory instructions between wait w and the previous wait / branch"""
        g, k = [], w - 1
        while k >= 0 and not insts[k]["op"].startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_endpgm")):
            if is_vmem(insts[k]["op"]):
                g.append(k)
            k -= 1
        return sorted(g)

    groups = {}
    for w in (w1, w2):
        g = group_before(w)
        if len(g) != n_loads or not all(is_stream(insts[k]) for k in g):
            bad.append(f"wait at {insts[w]['addr']:#x}: expected exactly {n_loads} stream loads before it, found "
                       f"{[insts[k]['op'] + ' ' + insts[k]['args'] for k in g]}")
        groups[w] = g
    if bad:
        return bad
    # the loop: first backward branch after w2 that lands before w1
    loop_end = loop_head = None
    for k in range(w2 + 1, len(insts)):
        t = insts[k]["target"]
        if insts[k]["op"].startswith(("s_cbranch", "s_branch")) and t is not None and t <= insts[w1]["addr"]:
            loop_end, loop_head = k, by_addr.get(t)
            break
    if loop_end is None or loop_head is None:
        return bad + ["no back edge found after the second counted wait"]
    dest = lambda k: vregs(insts[k]["args"].split(",")[0])
    d1 = set().union(*[dest(k) for k in groups[w1]])   # buffer filled before w1, covered by w2
    d2 = set().union(*[dest(k) for k in groups[w2]])   # buffer filled before w2, covered by w1 (next iteration)
    if d1 & d2:
        bad.append(f"the two buffers share registers {sorted(d1 & d2)}")

    def span_clean(load_idx, span, what):
        regs = dest(load_idx)
        for k in span:
            if k == load_idx:
                continue
            if vregs(insts[k]["args"]) & regs:
                bad.append(f"{what}: {insts[k]['op']} {insts[k]['args']} at {insts[k]['addr']:#x} touches "
                           f"v{sorted(regs)[0]}..v{sorted(regs)[-1]} while its load (at {insts[load_idx]['addr']:#x}) is in flight")
                return

    for k in groups[w1]:  # in flight from its issue to w2
        span_clean(k, range(k + 1, w2), "buffer 1")
    for k in groups[w2]:  # in flight from its issue, over the back edge, to w1
        span_clean(k, list(range(k + 1, loop_end + 1)) + list(range(loop_head, w1)), "buffer 0")
    # the prologue fills buffer 0 before the loop is entered
    pro = [k for k in range(0, loop_head) if is_stream(insts[k])]
    if len(pro) != n_loads or set().union(*[dest(k) for k in pro]) != d2:
        bad.append(f"prologue: expected {n_loads} stream loads into buffer 0 before the loop, found {len(pro)}")
    else:
        for k in pro:
            span_clean(k, list(range(k + 1, loop_head)) + list(range(loop_head, w1)), "prologue")
    # every way out of the loop drains the loads before anything else touches the buffers: the code behind the
    # loop starts (in address order) with trampolines back into the loop and the drain `s_waitcnt vmcnt(0)`
    both = d1 | d2
    drain = next((k for k in range(loop_end + 1, len(insts))
                  if insts[k]["op"] == "s_waitcnt" and insts[k]["args"].strip() == "vmcnt(0)"), None)
    if drain is None:
        return bad + ["no s_waitcnt vmcnt(0) behind the loop"]
    for k in range(loop_end + 1, drain):
        if vregs(insts[k]["args"]) & both:
            bad.append(f"{insts[k]['op']} {insts[k]['args']} at {insts[k]['addr']:#x} uses a stream buffer before the drain")

    def safe(t, depth=0):
        j = by_addr.get(t)
        if j is None or depth > 6:
            return False
        if loop_head <= j <= loop_end:
            return True
        if j > loop_end:   # behind the loop: at or before the drain, or a trampoline
            if j <= drain:
                return True
            return insts[j]["op"] == "s_branch" and safe(insts[j]["target"], depth + 1)
        # in front of the loop head: a block that falls into the loop (or leaves through safe branches)
        for q in range(j, loop_head):
            if vregs(insts[q]["args"]) & both:
                return False
            if insts[q]["op"].startswith(("s_cbranch", "s_branch")) and not safe(insts[q]["target"], depth + 1):
                return False
            if insts[q]["op"] == "s_branch":
                return True
        return True

    for k in range(loop_head, loop_end + 1):
        t = insts[k]["target"]
        if insts[k]["op"].startswith(("s_cbranch", "s_branch")) and t is not None and not safe(t):
            bad.append(f"branch at {insts[k]['addr']:#x} leaves the loop without reaching the drain s_waitcnt vmcnt(0)")
    return bad


def main(obj):
    funcs = functions(disassemble(obj))
    pat = re.compile(r"_ZN2gw14k_scan_cc_pipeILi(\d)ELb([01])ELb([01])EEEvNS_8ScanArgsE$")
    checked, failed = 0, 0
    for name, f in sorted(funcs.items()):
        m = pat.match(name)
        if not m:
            continue
        unroll, nt = int(m.group(1)), m.group(3) == "1"
        if not nt:
            continue  # without `nt` the stream loads cannot be told from the epilogue's loads; not a default shape
        problems = check_pipe(name, f["insts"], unroll, nt)
        checked += 1
        for p in problems:
            failed += 1
            print(f"check_isa: k_scan_cc_pipe<{unroll}, {m.group(2) == '1'}, true>: {p}", file=sys.stderr)
    if checked < 6:
        print(f"check_isa: only {checked} k_scan_cc_pipe instantiations found in {obj}", file=sys.stderr)
        return 1
    if failed:
        return 1
    print(f"check_isa: {checked} k_scan_cc_pipe instantiations keep their load pipeline intact")
    return 0


def unbundle(obj, d):
    fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.devnull], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET,
                    "--input=" + fat, "--output=" + co], check=True)
    return co


def kernel_metadata(co):
    """{mangled name: {key: int}} from the code object's AMDGPU metadata note"""
    text = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, {}
    for ln in text.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)\s*$", ln)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "agpr_count" and cur.get("name"):   # first key of the next kernel's map
            out[cur["name"]] = cur
            cur = {}
        cur[key] = val
    if cur.get("name"):
        out[cur["name"]] = cur
    return out


def main_no_scratch(obj, must_match=r"k_boost_mfma_fp4|k_boost_finish", at_least=9):
    with tempfile.TemporaryDirectory() as d:
        co = unbundle(obj, d)
        meta = kernel_metadata(co)
        funcs = functions(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True,
                                         text=True).stdout)
    checked, failed = 0, 0
    for name, m in sorted(meta.items()):
        if not re.search(must_match, name):
            continue
        checked += 1
        problems = []
        for key in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
            if int(m.get(key, "0")) != 0:
                problems.append(f"{key} = {m[key]}")
        n_scratch = sum(1 for i in funcs.get(name, {"insts": []})["insts"] if i["op"].startswith("scratch_"))
        if n_scratch:
            problems.append(f"{n_scratch} scratch_ instructions")
        if name not in funcs:
            problems.append("no ISA found for it")
        for p in problems:
            failed += 1
            print(f"check_isa: {name}: {p}", file=sys.stderr)
    if checked < at_least:
        print(f"check_isa: only {checked} kernels matching /{must_match}/ in {obj} (expected {at_least})", file=sys.stderr)
        return 1
    if failed:
        return 1
    print(f"check_isa: {checked} pair kernels of {os.path.basename(obj)} use no scratch memory")
    return 0


if __name__ == "__main__":
    if sys.argv[1] == "--no-scratch":   # --no-scratch obj [regex over the mangled names] [at least this many kernels]
        sys.exit(main_no_scratch(sys.argv[2], *(sys.argv[3:4] or [r"k_boost_mfma_fp4|k_boost_finish"]),
                                 *([int(sys.argv[4])] if len(sys.argv) > 4 else [9])))
    sys.exit(main(sys.argv[1]))
