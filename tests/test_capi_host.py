"""CPU-side checks of the product library (no compute calls): it loads, exports
every symbol include/gwas_scan.h declares, refuses to run without a GPU, and
its host logic (genotype-text packer, mask builder, geometry) agrees with the
oracle and with the reference's golden files."""
import os
import re

import numpy as np
import pytest

import libgwaspp_amd as gw
from libgwaspp_amd import capi
from golden_util import golden_sets, read_tped, read_tfam_status

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(gw.library_path()):
        gw.build_library()


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gwas_scan.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gw_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS)
    L = gw.library()
    for name in sorted(declared):
        assert hasattr(L, name), name


def test_no_silent_cpu_fallback():
    if gw.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(gw.GwError) as e:
        gw.GwTable(100, 10)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under libgwaspp_amd/ may
    import, link or open it."""
    pkg = os.path.join(ROOT, "libgwaspp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("liboracle", "gw_oracle", "from oracle", "import oracle", "gwo_"):
                    if needle in text:
                        # mentions inside comments naming the checker are fine; code use is not
                        for ln in text.splitlines():
                            if needle in ln:
                                s = ln.strip()
                                assert s.startswith(("//", "#", "*", "/*", '"""')) or "oracle/" in s, (f, ln)


def test_geometry_helpers(oracle):
    for n in (1, 15, 16, 17, 63, 64, 65, 1000, 10_000, 100_000, 500_000):
        assert gw.stream_blocks(n) == oracle.geometry_stream(n).data_blocks
        assert gw.block_blocks(n) == oracle.geometry_block(n).data_blocks


@pytest.mark.parametrize("name", golden_sets("plain") + golden_sets("cc"))
def test_packer_equals_oracle_on_golden_rows(oracle, name):
    n = int(name.split("_n")[1])
    for ln in read_tped(name):
        text = oracle.tped_genotype_text(ln)
        for layout, lname in ((gw.LAYOUT_STREAM, "stream"), (gw.LAYOUT_BLOCK, "block")):
            mine = gw.pack_genotype_row(text, n, layout)
            ref = oracle.pack_rows([text], n, lname)[0]
            assert np.array_equal(mine, ref)


def test_packer_first_seen_labelling_and_errors():
    # first homozygote seen gets code 1 whatever its alleles (quirk Q1)
    r = gw.pack_genotype_row(b"CC\tAC\tAA\t", 3)
    assert r[0] == (0x7000 | (5 << 8) | (1 << 4) | 0)
    nb = gw.stream_blocks(3)
    assert r[1] == 0b101 and r[1 + nb] == 0b110
    # heterozygote first, then homozygotes
    r = gw.pack_genotype_row(b"GT\tTT\tGG\t", 3)
    assert r[0] == (0x7000 | (15 << 8) | (11 << 4) | 10)
    for bad in (b"AC\tCA\t", b"AA\tCC\tGG\t", b"AA\tAC\tCC\tAG\t"):
        with pytest.raises(gw.GwError) as e:
            gw.pack_genotype_row(bad, 4)
        assert e.value.code == -6
    with pytest.raises(gw.GwError):
        gw.pack_genotype_row(b"AA\tAA\tAA\t", 2)  # more samples than columns
    # unknown symbols are missing calls (quirk Q2); the simulator's 'B' too
    r = gw.pack_genotype_row(b"AB\t00\tNN\tAA\t", 4)
    assert r[0] == 0x1000 and r[1] == 0b1000


@pytest.mark.parametrize("name", golden_sets("cc")[:12])
def test_mask_builder_equals_oracle(oracle, name):
    n = int(name.split("_n")[1])
    cases, ctrls = read_tfam_status(name)
    for idx in (cases, ctrls):
        m, cnt = gw.build_stream_mask(idx, n)
        _, om, ocnt = oracle.ccs_masks(idx, n)
        assert cnt == ocnt and np.array_equal(m, om)


def test_mask_builder_edges():
    m, cnt = gw.build_stream_mask([3, 3, 5], 10)
    assert cnt == 2 and m[0] == 0b101000
    with pytest.raises(gw.GwError) as e:
        gw.build_stream_mask([10], 10)
    assert e.value.code == -5
    m, cnt = gw.build_stream_mask([], 10)
    assert cnt == 0 and not m.any()


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/gwas_scan.h compiles as C99 and the library links from a C program
    (examples/scan_demo.c); its GPU-free half runs here."""
    import subprocess
    exe = str(tmp_path / "scan_demo")
    pkg = os.path.join(ROOT, "libgwaspp_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "scan_demo.c"), "-L" + pkg, "-lgwscan", "-Wl,-rpath," + pkg,
                    "-o", exe], check=True)
    out = subprocess.run([exe, "--host-only"], check=True, capture_output=True, text=True).stdout
    assert "row 0 header 0x7015" in out          # AA first, then AC, then CC
    assert "case mask 0x0055 (4)" in out and "control mask 0x00aa (4)" in out


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py has no CPU path to fall back to: without a GPU it must end non-zero and say why (it is the HIP
    scan that is measured, never the oracle)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("this machine has a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "no GPU visible" in r.stderr and not r.stdout.strip()
    h = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert h.returncode == 0 and "--gpus" in h.stdout and "--workload" in h.stdout
