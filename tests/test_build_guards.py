"""Build-time guards (CPU only).

1. Every pure virtual of the reference's table interface (geno_table.h:51-65,
   single_marker_analyzable.h:132-145, pairwise_marker_analyzable.h:175-187, case_control_selectable.h:42,
   util/table/table.h:69 and the virtual operator() at :53) has an identically typed `override` in
   libgwaspp_amd/host/include/genetics/genotype/hip_genotype_table.h -- parsed from the reference's own
   headers, so it runs only where /root/reference exists (never on the GPU box).
2. host/src/hip_genotype_table.cpp compiles UNCHANGED against the reference's own headers (reference include
   directories in front of this repo's mirror): `g++ -std=gnu++14 -fsyntax-only`.  The reference's
   common_genotype_func.h:36-37 includes Rmath.h, which this image lacks; for this SYNTAX check only, a
   one-declaration Rmath.h is written to a temporary directory (it pins nothing and is no oracle).
3. The ISA guard of the pipelined scan (libgwaspp_amd/csrc/check_isa.py): passes on the object the
   Makefile built and catches a tampered pipeline."""
import copy
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
HIP_H = os.path.join(ROOT, "libgwaspp_amd", "host", "include", "genetics", "genotype", "hip_genotype_table.h")
CSRC = os.path.join(ROOT, "libgwaspp_amd", "csrc")

BUILTIN = {"int", "char", "uint", "ushort", "ulong", "bool", "void", "double", "float", "long", "unsigned",
           "string", "DataBlock"}


def _norm_param(p):
    toks = re.findall(r"[A-Za-z_][\w:]*|[&*]", p.replace("std::", ""))
    if len(toks) >= 2 and toks[-1] not in ("&", "*") and toks[-1] not in BUILTIN and toks[-1] != "const":
        toks = toks[:-1]          # the parameter's name
    return " ".join(toks)


def _norm(ret, name, params):
    ps = [_norm_param(p) for p in params.split(",") if p.strip()]
    ret = " ".join(re.findall(r"[A-Za-z_][\w:]*|[&*]", ret.replace("std::", "").replace("virtual", "")))
    return f"{ret} {name}({', '.join(ps)})"


def _declarations(text, want):
    """want='pure': `virtual R f(...) = 0;`   want='override': `R f(...) override;`"""
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = set()
    if want == "pure":
        pat = re.compile(r"virtual\s+([^;(){}]+?)\s*(\boperator\s*\(\s*\)|\b\w+)\s*\(([^()]*)\)\s*(?:const\s*)?=\s*0\s*;", re.S)
        for m in pat.finditer(text):
            out.add(_norm(m.group(1), re.sub(r"\s+", "", m.group(2)), m.group(3)))
        return out
    pat = re.compile(r"^(.+?)\s*(\boperator\s*\(\s*\)|\b\w+)\s*\(([^()]*)\)\s*(?:const\s*)?override$", re.S)
    for stmt in re.split(r"[;{}]", text):
        stmt = re.sub(r"^\s*(?:public|protected|private)\s*:", "", stmt.strip()).strip()
        m = pat.match(stmt)
        if m:
            out.add(_norm(m.group(1), re.sub(r"\s+", "", m.group(2)), m.group(3)))
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
def test_every_reference_pure_virtual_has_an_identically_typed_override():
    g = os.path.join(REF, "libgwaspp", "genetics", "genotype")
    pure = set()
    for path in (os.path.join(g, "geno_table.h"), os.path.join(g, "single_marker_analyzable.h"),
                 os.path.join(g, "pairwise_marker_analyzable.h"), os.path.join(g, "case_control_selectable.h"),
                 os.path.join(REF, "util", "table", "table.h")):
        with open(path) as f:
            pure |= _declarations(f.read(), "pure")
    with open(HIP_H) as f:
        over = _declarations(f.read(), "override")
    assert len(pure) == 19, sorted(pure)     # 6 + 5 + 6 + 1 + Table::initialize(), counted from the headers
    missing = sorted(pure - over)
    assert not missing, f"no identically typed override in hip_genotype_table.h for: {missing}"
    # operator()(r, c) is virtual with a body in the reference (table.h:53): overridden with the same type
    assert "DataBlock operator()(int, int)" in over or "D operator()(int, int)" in over


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
def test_subclass_compiles_unchanged_against_the_reference_headers(tmp_path):
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    (tmp_path / "Rmath.h").write_text("double pchisq(double, double, int, int);\n")
    host = os.path.join(ROOT, "libgwaspp_amd", "host")
    cmd = ["g++", "-std=gnu++14", "-fsyntax-only", "-Wall", "-H",
           "-I" + REF, "-I" + os.path.join(REF, "libgwaspp"), "-I" + str(tmp_path),
           "-I" + os.path.join(host, "include"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(host, "src", "hip_genotype_table.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    assert " error: " not in r.stderr
    # -H lists every header opened: the table interface and CaseControlSet came from the reference tree,
    # only hip_genotype_table.h itself from this repo's mirror directory
    opened = [ln.lstrip(". ").strip() for ln in r.stderr.splitlines() if ln.startswith(".")]
    for name in ("genetics/genotype/geno_table.h", "genetics/analyzable/case_control_set.h",
                 "genetics/genotype/single_marker_analyzable.h", "genetics/genotype/pairwise_marker_analyzable.h",
                 "genetics/genotype/case_control_selectable.h", "util/table/table.h"):
        hits = [h for h in opened if h.endswith(name)]
        assert hits and all(h.startswith(REF) for h in hits), (name, hits)
    mirror = [h for h in opened if h.startswith(os.path.join(host, "include"))]
    assert [os.path.basename(h) for h in mirror] == ["hip_genotype_table.h"], mirror


def test_mirror_case_control_set_offers_nothing_the_reference_class_lacks():
    """The subclass may only use what analyzable/case_control_set.h:48-91 declares; the mirror keeps no extras
    that could creep back in (round 2: revision(), stream_block_count())."""
    with open(os.path.join(ROOT, "libgwaspp_amd", "host", "include", "genetics", "analyzable",
                           "case_control_set.h")) as f:
        text = f.read()
    public = text[text.index("public:"):text.index("protected:")]
    names = set(re.findall(r"(\w+)\s*\(", public))
    allowed = {"CaseControlSet", "getPossibleIndices", "getMaximumIndex", "getCaseCount", "getControlCount",
               "getTotalCount", "setCases", "setControls", "setAllAsCases", "setAllAsControls", "control_begin",
               "control_end", "case_begin", "case_end", "stream_case_begin", "stream_case_end",
               "stream_control_begin", "stream_control_end", "isCase", "isControl", "reset",
               "data", "size"}      # data()/size(): std::vector calls inside the inline bodies
    assert names <= allowed, sorted(names - allowed)


def _check_isa():
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(CSRC, "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def pipe_functions():
    obj = os.path.join(CSRC, "build", "gw_kernels.o")
    if not os.path.exists(obj):
        pytest.skip("libgwaspp_amd/csrc/build/gw_kernels.o not built here")
    ci = _check_isa()
    return ci, ci.functions(ci.disassemble(obj))


def test_isa_guard_passes_on_the_built_kernels(pipe_functions):
    ci, funcs = pipe_functions
    name = "_ZN2gw14k_scan_cc_pipeILi2ELb1ELb1EEEvNS_8ScanArgsE"      # the c3 default
    assert name in funcs
    assert ci.check_pipe(name, funcs[name]["insts"], 2, True) == []
    assert ci.main(os.path.join(CSRC, "build", "gw_kernels.o")) == 0


def test_isa_guard_catches_a_broken_pipeline(pipe_functions):
    ci, funcs = pipe_functions
    name = "_ZN2gw14k_scan_cc_pipeILi2ELb1ELb1EEEvNS_8ScanArgsE"
    insts = funcs[name]["insts"]
    waits = [k for k, i in enumerate(insts) if i["op"] == "s_waitcnt" and i["args"].strip() == "vmcnt(4)"]
    assert len(waits) == 2
    loads = [k for k in range(waits[0] - 20, waits[0]) if insts[k]["op"] == "global_load_dwordx4"]
    first_dest = insts[loads[0]]["args"].split(",")[0].strip()        # e.g. v[32:35]
    reg = "v" + re.search(r"\d+", first_dest).group(0)

    # (1) the compiler moves a buffer register while its load is in flight
    t = copy.deepcopy(insts)
    t.insert(waits[0] + 3, {"op": "v_mov_b32_e32", "args": f"v200, {reg}", "addr": t[waits[0] + 3]["addr"] - 1, "target": None})
    assert any("in flight" in p for p in ci.check_pipe(name, t, 2, True))
    # (2) a load of the group disappears (or is sunk behind the wait)
    t = copy.deepcopy(insts)
    del t[loads[-1]]
    assert any("expected exactly 4 stream loads" in p for p in ci.check_pipe(name, t, 2, True))
    # (3) another vector-memory instruction lands inside the group
    t = copy.deepcopy(insts)
    t.insert(loads[-1], {"op": "global_store_dword", "args": "v[0:1], v2, off", "addr": t[loads[-1]]["addr"] - 1, "target": None})
    assert any("expected exactly 4 stream loads" in p for p in ci.check_pipe(name, t, 2, True))
    # (4) a spill
    t = copy.deepcopy(insts)
    t.insert(5, {"op": "scratch_store_dword", "args": "off, v1, s0", "addr": t[5]["addr"] - 1, "target": None})
    assert any("scratch" in p for p in ci.check_pipe(name, t, 2, True))
    # (5) the counted wait is weakened
    t = copy.deepcopy(insts)
    t[waits[1]]["args"] = "vmcnt(5)"
    assert ci.check_pipe(name, t, 2, True)
