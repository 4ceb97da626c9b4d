#!/usr/bin/env python3
"""Diagnostic: writes patched copies of csrc/covariance.hpp (one piece of k_covariance_tiled removed each) and builds one probe binary per
copy, to see which piece its duration follows.  Results are meaningless numerically; only the times matter."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "coherent-rtlsdr_amd/csrc/covariance.hpp")).read()
out = os.path.join(root, "tools/cov_probe/_build"); os.makedirs(out, exist_ok=True)
def no_mfma(s):
    return re.sub(r"(g[123][ab]) = __builtin_amdgcn_mfma_i32_32x32x32_i8\((\w+), (\w+), g[123][ab], 0, 0, 0\);", r"\1[0] ^= \2.x ^ \3.y;", s)
def no_loop_loads(s):
    return s.replace("if (c + 2 < c_hi) COV_GLOAD(p0, c + 2);", "").replace("if (c + 3 < c_hi) COV_GLOAD(p1, c + 3);", "")
def no_stores(s):
    return s.replace("    int *pt = partial + (size_t)item * 2 * CT * CT;", "    int *pt = partial + (size_t)item * 2 * CT * CT;\n    if (g1a[0] != 0x12345678) return;")
def no_lds_reads(s):
    s = s.replace("const v4i a0v = *reinterpret_cast<const v4i *>(Ab + ks * 32), a1v = *reinterpret_cast<const v4i *>(Ab + 32 * CPITCH + ks * 32);",
                  "const v4i a0v = v4i{ks, lane, 3, 4}, a1v = v4i{lane, ks, 5, 6};")
    return s.replace("const v4i bv = *reinterpret_cast<const v4i *>(Bb + ks * 32);", "const v4i bv = v4i{7, ks, lane, 9};")
variants = {"as_is": lambda s: s, "no_mfma": no_mfma, "no_loop_loads": no_loop_loads, "no_stores": no_stores, "no_lds_reads": no_lds_reads,
            "mfma_only": lambda s: no_stores(no_loop_loads(no_lds_reads(s)))}
for name, f in variants.items():
    t = f(src)
    assert name == "as_is" or t != src, name
    hp = os.path.join(out, name + ".hpp"); open(hp, "w").write(t)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f'-DCOV_HEADER="{hp}"', f'-DCOV_NAME="{name}"',
                           os.path.join(root, "tools/cov_probe/probe.hip"), "-o", os.path.join(out, name)])
    print("built", name)
