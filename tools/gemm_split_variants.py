#!/usr/bin/env python3
"""Timing-only variants of k_s2s_gemm_split (results are garbage): which part of a k step costs what.

  gemm_split_variants.py build   -> aether_amd/libaether_gsvar{N}.so from patched scratch copies of csrc/
  on the GPU box:                   tools/gemm_split_variants_run.sh [rollout args]      (add  --opt gemm_split=3  to put every
                                    launch on this kernel: by default launches of > 256 workgroups use k_s2s_gemm_split_r1)
  results of round 4:               profiles/r04_seq2seq_gemm_split_variants.txt

  1: X by LDS-DMA from CONTIGUOUS addresses (1 KB per instruction instead of 16 rows x 64 B)
  2: no X DMA at all   3: no MFMAs   4: no weight DMA   5: no workgroup barrier per step
  6: one k step per tile   7: no epilogue
"""
import os, re, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from aether_amd import build as B

def sub(path, old, new, count=1):
    s = open(path).read()
    assert s.count(old) == count, (path, old, s.count(old))
    open(path, "w").write(s.replace(old, new))

def kernel_span(s):
    a = s.index("k_s2s_gemm_split(const S2SJobs jobs) {")
    b = s.index("// (A/B: the rounds 2 - 4 structure")
    return a, b

def patch(v, root):
    f = os.path.join(root, "aether_amd", "csrc", "s2s_step.h")
    s = open(f).read()
    a, b = kernel_span(s)
    k = s[a:b]
    if v == 1:
        old = "                __builtin_amdgcn_global_load_lds(p + 16 * hf,"
        assert k.count(old) == 1
        k = k.replace(old, "                __builtin_amdgcn_global_load_lds(J.X + (size_t)(blockIdx.x & 63) * 8192 + (s & 7) * 1024 + (2 * nb + hf) * 256 + 4 * lane + 0 * (p - J.X),")
    if v == 2:
        old = k[k.index("        unsigned char* xd = xring + slot * XSTAGE;"):k.index("    };\n    // this lane's two chunks")]
        k = k.replace(old, "")
        k = k.replace("constexpr int NDMA = 4 + 2 * NB;", "constexpr int NDMA = 4;")
    if v == 3:
        k, n = re.subn(r"__builtin_amdgcn_mfma_f32_16x16x32_f16\((w[lh]), (x[hl]\[nb\]), acc\[mb\]\[nb\], 0, 0, 0\)", r"acc[mb][nb]", k)
        assert n == 3, n
    if v == 4:
        old = k[k.index("#pragma unroll\n        for (int f = 0; f < 4; ++f) {\n            const int fr = wave + 4 * f;"):k.index("        unsigned char* xd = xring + slot * XSTAGE;")]
        k = k.replace(old, "")
        k = k.replace("constexpr int NDMA = 4 + 2 * NB;", "constexpr int NDMA = 2 * NB;")
    if v == 5:
        old = "        lds_barrier();                                        // weight fragments of step s visible to every wave; slot (s - 1) % NST is free\n"
        assert k.count(old) == 1
        k = k.replace(old, "")
    if v == 6:      # one k step per tile: what a tile costs besides its steps
        old = "    const int s1 = J.K >> 5, s2 = J.W2img != nullptr ? J.K2 >> 5 : 0, S = s1 + s2;"
        assert k.count(old) == 1
        k = k.replace(old, "    const int s1 = 1, s2 = 0, S = 1;")
    if v == 7:      # no epilogue (one store per lane keeps the accumulators alive)
        old = k[k.index("    const int act = J.act;"):]
        k = k.replace(old, "    { float t = 0.f;\n#pragma unroll\n      for (int mb = 0; mb < 8; ++mb)\n#pragma unroll\n        for (int nb = 0; nb < NB; ++nb) t += (acc[mb][nb][0] + acc[mb][nb][1] + acc[mb][nb][2] + acc[mb][nb][3]) * inv_xs[nb];\n      if (n0 + i < N) Y[(size_t)(n0 + i) * ldy + m0 + 4 * q] = t; }\n}\n\n")
    open(f, "w").write(s[:a] + k + s[b:])

VARIANTS = tuple(int(x) for x in os.environ.get('GSVARS', '1,2,3,4,5,6,7').split(','))

def build():
    for v in VARIANTS:
        root = os.path.join(REPO, "build", "gs_diag_%d" % v)
        shutil.rmtree(root, ignore_errors=True)
        os.makedirs(root)
        shutil.copytree(os.path.join(REPO, "aether_amd", "csrc"), os.path.join(root, "aether_amd", "csrc"))
        shutil.copytree(os.path.join(REPO, "include"), os.path.join(root, "include"))
        patch(v, root)
        out = os.path.join(REPO, "aether_amd", "libaether_gsvar%d.so" % v)
        cmd = [B.hipcc_path(), *B.FLAGS, os.path.join(root, "aether_amd", "csrc", "aether_hip.hip"), "-o", out]
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build()
    else:
        print(__doc__)
