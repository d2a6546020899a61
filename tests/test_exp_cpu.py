"""csrc/hf_exp.h on the CPU: the restatement of glibc's exp that the emission kernels run against the host's libm,
bit for bit, and the generated table against its formula."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "%s/flagger_amd/csrc/hf_exp.h"
#include <cstdio>
#include <random>
int main() {
    std::mt19937_64 g(2024);
    long bad = 0, n = 0;
    auto test = [&](double x) {
        const double a = hf_exp(x), b = std::exp(x); n++;
        if (hf_exp_bits(a) != hf_exp_bits(b) && !(a != a && b != b)) { if (bad < 10) std::printf("x=%%a mine=%%a libm=%%a\n", x, a, b); bad++; }
    };
    std::uniform_real_distribution<double> u(-760.0, 720.0), v(-60.0, 0.0), w(-2e-3, 2e-3);
    for (long i = 0; i < 4000000; i++) { test(u(g)); test(v(g)); test(w(g)); }
    for (long i = 0; i < 1000000; i++) { uint64_t b = g(); double x; std::memcpy(&x, &b, 8); test(x); }
    const double sp[] = {0.0, -0.0, 709.78, 709.79, -708.4, -745.13, -745.14, -1074.0, 1e-300, -1e-300, INFINITY, -INFINITY, 512.0, -512.0,
                         1024.0, -1024.0, 0x1p-54, -0x1p-54, 0x1p-55};
    for (double x : sp) test(x);
    std::printf("%%ld arguments, %%ld differ\n", n, bad);
    return bad != 0;
}
"""


def _has_fma():
    try:
        return "fma" in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split()
    except OSError:
        return False


@pytest.mark.skipif(not _has_fma(), reason="glibc runs its non-FMA build of exp on this CPU: other roundings by design")
def test_hf_exp_equals_libm_exp_bit_for_bit(tmp_path):
    src = tmp_path / "exp_check.cpp"
    src.write_text(SRC % ROOT)
    exe = tmp_path / "exp_check"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "13000019 arguments, 0 differ" in r.stdout, r.stdout


def test_exp_table_is_what_its_formula_says():
    """2^(k/128) = H[k] (1 + T[k]): the committed header against mpmath (profiles/tools/gen_exp_table.py wrote it the same way)."""
    import re
    import struct
    import mpmath as mp
    mp.mp.prec = 300
    txt = open(os.path.join(ROOT, "flagger_amd", "csrc", "hf_exp_table.h")).read()
    words = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ull", txt)]
    assert len(words) == 256
    for k in range(128):
        v = mp.power(2, mp.mpf(k) / 128)
        H = float(v)
        T = float((v - mp.mpf(H)) / mp.mpf(H))
        assert words[2 * k] == struct.unpack("<Q", struct.pack("<d", T))[0]
        assert words[2 * k + 1] == struct.unpack("<Q", struct.pack("<d", H))[0] - ((k << 52) // 128)
