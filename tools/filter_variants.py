#!/usr/bin/env python3
"""Timing-only variants of k_s2s_filter_split (results are garbage): which part of a GEMM step costs what.

  filter_variants.py build        -> aether_amd/libaether_filtvar{1..5}.so from patched scratch copies of csrc/
  then on the GPU box:  python tools/_alt_lib_run.py libaether_filtvar<N>.so tools/s2s_rollout_only.py  under rocprofv3

  1: no MFMAs   2: no workgroup barrier per step   3: no LDS-DMA in the step loop (and no wait for it)
  4: no fragment reads in the step loop   5: no weighting FMAs (out += ea * Z)
"""
import os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from aether_amd import build as B

def sub(path, old, new, count=1):
    s = open(path).read()
    assert s.count(old) == count, (path, old, s.count(old))
    open(path, "w").write(s.replace(old, new))

def patch(v, root):
    f = os.path.join(root, "aether_amd", "csrc", "s2s_filter.h")
    if v == 1:
        import re
        s = open(f).read()
        s, n = re.subn(r"__builtin_amdgcn_mfma_f32_16x16x32_f16\(([^;]*?), (kb == 0 \? f32x4\{0\.f, 0\.f, 0\.f, 0\.f\} : tmp\[nb\]|tmp\[nb\]), 0, 0, 0\)",
                       r"(\2)", s)
        assert n == 3, n
        open(f, "w").write(s)
    if v == 2:
        sub(f, "            lds_barrier();                                     // ... for every wave\n", "")
    if v == 3:
        sub(f, "                dma_next();                                    // step it + NST -> slot it % NST\n", "")
        sub(f, '                    if (it + 2 < IT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");\n'
               '                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n', "")
    if v == 4:
        sub(f, "                if (more) issue_reads(wn, en, slot_n, r_n);\n", "", count=2)
    if v == 5:
        sub(f, "                for (int nb = 0; nb < 4; ++nb) outv[mb][nb] += tmp[nb] * own_reg(es[nb]);       // out += ea[:, r] * Z_r\n",
            '                for (int nb = 0; nb < 4; ++nb) asm volatile("" :: "v"(tmp[nb]), "v"(es));\n')
    if v == 6:      # both groups in the same order (lock step, as before the anti-phase change)
        sub(f, "    const bool late = chalf != 0; ", "    const bool late = false; (void)chalf; // ")


VARIANTS = (1, 2, 3, 4, 5)


def build():
    for v in VARIANTS:
        root = os.path.join(REPO, "build", "filt_diag_%d" % v)
        shutil.rmtree(root, ignore_errors=True)
        os.makedirs(root)
        shutil.copytree(os.path.join(REPO, "aether_amd", "csrc"), os.path.join(root, "aether_amd", "csrc"))
        shutil.copytree(os.path.join(REPO, "include"), os.path.join(root, "include"))
        patch(v, root)
        out = os.path.join(REPO, "aether_amd", "libaether_filtvar%d.so" % v)
        cmd = [B.hipcc_path(), *B.FLAGS, os.path.join(root, "aether_amd", "csrc", "aether_hip.hip"), "-o", out]
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

if __name__ == "__main__":
    build()
