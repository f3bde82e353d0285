"""Host-side check of device arithmetic that has a plain-C++ statement in the same header:
mk_device.hpp's 32-bit-halves fingerprint (used by the build's hash loop) against its 64-bit
statement of Miekki::mantis, over all h, both widths, random and edge operands."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mantis_in_32bit_halves_equals_the_64bit_statement(tmp_path):
    src = open(os.path.join(ROOT, "miekki_amd", "csrc", "mk_device.hpp")).read()
    hdr = tmp_path / "mk_device_host.hpp"
    hdr.write_text(re.sub(r"#include <hip/hip_runtime.h>", "", src))
    exe = tmp_path / "mantis_check"
    subprocess.run(["g++", "-O2", "-std=c++17", f'-DMK_DEVICE_HPP="{hdr}"', "-o", str(exe),
                    os.path.join(ROOT, "tests", "helpers", "mantis_check.cpp")], check=True)
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "mismatches 0" in out
