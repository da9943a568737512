"""Host-side C++ of the product under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "build")


def test_ingest_logic_under_asan_ubsan():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "pack_fuzz")
    csrc = os.path.join(ROOT, "libgwaspp_amd", "csrc")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I" + csrc, "-o", exe, os.path.join(ROOT, "tests", "cpp", "pack_fuzz.cpp"),
                    os.path.join(csrc, "gw_pack.cpp")], check=True)
    r = subprocess.run([exe, "3000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "accepted" in r.stdout
