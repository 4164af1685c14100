#!/usr/bin/env python3
"""Attribute the rare-tile corruption of round 2 to an instruction pattern.

k_edge_layer1 compiled WITHOUT its register cap (accumulators in AGPRs) produced ~3 wrong 16-edge tiles per million.
The uncapped build differs from the product build in two patterns (tools/isa_check.py): R1, LDS loads that land in
accumulator registers and feed a bf16 MFMA's SrcC, and R2, bf16 MFMAs whose destination tuple partially overlaps the
SrcC tuple.  The diagnostic variants below form a 2 x 2 design.  They are NOT in the product sources (round 4): build()
copies csrc/ to a scratch directory and patches its own copy of k_edge_layer1 / edge_features there (PATCHES):

    variant 0: R1 + R2 (as compiled in round 2)      variant 1: R2 only        variant 2: R1 only
    variant 3: R1 with a full wait + 8 idle states    variant 4: neither (accumulators still in AGPRs)
    variants 5-8: tile order reversed / full waits at three places (where does "always the batch's last tile" come from?)
    variant 9: as 0, the frame sums of the feature build kept out of packed FMAs      variant 10: as 0, product flags

  hazard_variants.py build            -> aether_amd/libaether_hip_haz{0..4}.so (hipcc, here or on the GPU box)
  hazard_variants.py run ref|0..4 [reps]
        ref = the product library; writes gpurun_out/haz_ref_hash.pt (per-tile hashes of e1 at B=32, N=1024);
        a variant is compared tile by tile against that file (same arithmetic in the same order: must be bit-equal)
        and against its own first repetition.
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


# ---- the diagnostic variants as source patches of a scratch copy of csrc/ (the product compiles ONE k_edge_layer1)
_AFTER_LOAD = {
    1: 'asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]));',                 # through VGPRs (no load lands in an AGPR)
    2: 'asm volatile("" : "+a"(A[0]), "+a"(A[1]), "+a"(A[2]), "+a"(A[3]));',                 # loads land in AGPRs, tuples pinned whole
    3: 'asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_nop 7" : "+a"(A[0]), "+a"(A[1]), "+a"(A[2]), "+a"(A[3]));',   # + full wait, 8 idle states
    4: 'asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3])); asm volatile("" : "+a"(A[0]), "+a"(A[1]), "+a"(A[2]), "+a"(A[3]));',
}
_AFTER_GEMM = {v: 'asm volatile("" : "+a"(A[0]), "+a"(A[1]), "+a"(A[2]), "+a"(A[3]));' for v in (2, 3, 4)}


def _patch_sources(v, root):
    """Apply variant v to the copy of csrc/ under root.  Every anchor has to be found exactly once."""
    def sub(path, old, new):
        s = open(path).read()
        assert s.count(old) == 1, (path, old)
        open(path, "w").write(s.replace(old, new))
    st = os.path.join(root, "csrc", "streamed.h")
    sub(st, "__global__ void __launch_bounds__(256, 2)\nk_edge_layer1(", "__global__ void __launch_bounds__(256)\nk_edge_layer1(")
    if v in _AFTER_LOAD:
        sub(st, "                gemm_split<4, 1>(w1, bop[t], acc, lane);\n",
            "                { auto& A = acc; %s }\n                { auto& A = acc2; %s }\n"
            "                gemm_split<4, 1>(w1, bop[t], acc, lane);\n%s" %
            (_AFTER_LOAD[v], _AFTER_LOAD[v], ("                { auto& A = acc; %s }\n" % _AFTER_GEMM[v]) if v in _AFTER_GEMM else ""))
        if v in _AFTER_GEMM:
            sub(st, "                gemm_split<4, 2>(w2, h1, acc2, lane);\n",
                "                gemm_split<4, 2>(w2, h1, acc2, lane);\n                { auto& A = acc2; %s }\n" % _AFTER_GEMM[v])
    if v == 5:      # tiles of a batch in reverse order: does the failure follow the position or the rows?
        sub(st, "        for (int t = 0; t < 4; ++t) {\n            const int64_t k = batch * 64 + 16 * t + i;",
            "        for (int tt = 0; tt < 4; ++tt) {\n            const int t = 3 - tt;\n            const int64_t k = batch * 64 + 16 * t + i;")
    if v == 6:      # everything of a batch complete before the next batch's features are built
        sub(st, "                tile_receiver_sums(eo, wfeat, gsel[tile * 64 + lane], rcv, tile, part, i, q, lane);\n            }\n        }\n",
            "                tile_receiver_sums(eo, wfeat, gsel[tile * 64 + lane], rcv, tile, part, i, q, lane);\n            }\n        }\n"
            '        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");\n')
    if v == 7:      # feature rows complete in LDS before any lane reads another lane's row
        sub(st, "        __builtin_amdgcn_wave_barrier();\n        f32x4 bop[4][2];\n",
            '        __builtin_amdgcn_wave_barrier();\n        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");\n        f32x4 bop[4][2];\n')
    if v == 8:      # all eight operand reads back before the first tile starts
        sub(st, "        __builtin_amdgcn_wave_barrier();    // features are in registers: the rows become tile staging\n",
            "        __builtin_amdgcn_wave_barrier();    // features are in registers: the rows become tile staging\n"
            '        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");\n')
    if v == 9:      # the frame sums of the feature build kept out of packed FMAs
        sub(os.path.join(root, "csrc", "common.h"), "            s2 += rba * nj[NI::F + b];\n",
            '            s2 += rba * nj[NI::F + b];\n            asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2));\n')


def build(variants=range(5)):
    import shutil
    import tempfile
    from aether_amd import build as B
    procs = []
    scratch = tempfile.mkdtemp(prefix="aether_haz_")
    for v in variants:
        root = os.path.join(scratch, f"v{v}", "aether_amd")
        shutil.copytree(os.path.join(REPO, "aether_amd", "csrc"), os.path.join(root, "csrc"))
        shutil.copytree(os.path.join(REPO, "include"), os.path.join(scratch, f"v{v}", "include"))
        _patch_sources(v, root)
        out = os.path.join(REPO, "aether_amd", f"libaether_hip_haz{v}.so")
        # variants 0-9 reproduce round 2's build: SLP vectoriser on (it is what forms the packed FMAs with op_sel);
        # variant 10 is the uncapped kernel under the product's flags (build.py: -fno-slp-vectorize) -- the fix
        flags = [f for f in B.FLAGS if f != "-fno-slp-vectorize"] if v < 10 else list(B.FLAGS)
        cmd = [B.hipcc_path(), *flags, os.path.join(root, "csrc", "aether_hip.hip"), "-o", out]
        procs.append((v, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    for v, p in procs:
        assert p.wait() == 0, v
        print("built variant", v, flush=True)
    shutil.rmtree(scratch, ignore_errors=True)


def run(which, reps):
    import torch
    from aether_amd import _lib
    if which != "ref":
        _lib.LIB_PATH = os.path.join(REPO, "aether_amd", f"libaether_hip_haz{which}.so")
    from aether_amd.nn.state2state.aether import Aether
    from aether_amd.synthetic import make_batch
    torch.manual_seed(1)
    m = Aether(4, 64, 0.0, 2, device="cuda")
    m.flags = _lib.FLAG_FORCE_STREAMED | _lib.FLAG_KEEP_INTERMEDIATES
    B, N = 32, 1024
    inp = make_batch(B, N, 2, seed=3, device="cuda")
    Nn, E = B * N, inp["edges"][0].numel()
    path = os.path.join(REPO, "gpurun_out", "haz_ref_hash.pt")
    ref = torch.load(path).cuda() if which != "ref" and os.path.exists(path) else None
    first = None
    tot_ref = tot_self = 0
    for rep in range(reps):
        with torch.no_grad():
            m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
        e1 = m.debug_fetch("e1", Nn, E, 64)
        h = e1.view(torch.int32).view(-1, 16 * 64).to(torch.int64).sum(1)
        if first is None:
            first = h.clone()
        tot_self += int((h != first).sum())
        if ref is not None:
            tot_ref += int((h != ref).sum())
        del e1, h
    tiles = first.numel()
    if which == "ref":
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(first.cpu(), path)
    print(f"variant {which}: {reps} repetitions x {tiles} tiles; tiles differing from the product build's: "
          f"{tot_ref if ref is not None else 'n/a'}; from this build's first repetition: {tot_self}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build([int(a) for a in sys.argv[2:]] or range(5))
    elif sys.argv[1] in ("sig", "featsig", "forensic"):
        pass
    else:
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 10)


def signature(which, reps):
    """Where inside a tile do the wrong values sit?  Column c = 16 mb + 4 q + r of row i is register r of accumulator
    tuple mb in lane (i, q): the pattern of wrong entries names the registers and lanes that were corrupted."""
    import torch
    from aether_amd import _lib
    if which != "ref":
        _lib.LIB_PATH = os.path.join(REPO, "aether_amd", f"libaether_hip_haz{which}.so")
    from aether_amd.nn.state2state.aether import Aether
    from aether_amd.synthetic import make_batch
    torch.manual_seed(1)
    m = Aether(4, 64, 0.0, 2, device="cuda")
    m.flags = _lib.FLAG_FORCE_STREAMED | _lib.FLAG_KEEP_INTERMEDIATES
    B, N = 32, 1024
    inp = make_batch(B, N, 2, seed=3, device="cuda")
    Nn, E = B * N, inp["edges"][0].numel()

    def fwd():
        with torch.no_grad():
            m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
        return m.debug_fetch("e1", Nn, E, 64).view(-1, 16, 64)

    a = fwd().clone()
    b = fwd().clone()
    # a good copy: where a and b differ take a third opinion
    good = a.clone()
    d = (a != b).flatten(1).any(1).nonzero().flatten()
    if d.numel():
        c = fwd()
        for t in d.tolist():
            good[t] = b[t] if torch.equal(b[t], c[t]) else a[t]
    del a, b
    shown = 0
    stats = {"tiles": 0, "t%4": {}, "mb": {}, "r": {}, "q": {}, "rows": {}, "stale": 0}
    for rep in range(reps):
        e = fwd()
        bad = (e != good).flatten(1).any(1).nonzero().flatten().tolist()
        for t in bad:
            diff = (e[t] != good[t])                                  # [16 rows][64 cols]
            cols = diff.any(0).nonzero().flatten().tolist()
            rows = diff.any(1).nonzero().flatten().tolist()
            mbs = sorted({c // 16 for c in cols}); qs = sorted({(c % 16) // 4 for c in cols}); rs = sorted({c % 4 for c in cols})
            stats["tiles"] += 1
            for k, v in (("t%4", t % 4), ("mb", tuple(mbs)), ("q", tuple(qs)), ("r", tuple(rs)), ("rows", len(rows))):
                stats[k][v] = stats[k].get(v, 0) + 1
            # does the wrong block equal the right value of a neighbouring tile (a stale or misdirected register)?
            stale = None
            for dt in (-1, 1, -2, 2, -3, 3, -4, 4, -8, 8, -16, 16):
                u = t + dt
                if 0 <= u < e.shape[0] and all(torch.equal(e[t][:, 16 * mb:16 * mb + 16][diff[:, 16 * mb:16 * mb + 16]],
                                                           good[u][:, 16 * mb:16 * mb + 16][diff[:, 16 * mb:16 * mb + 16]]) for mb in mbs):
                    stale = dt
                    break
            stats["stale"] += stale is not None
            if shown < 3:
                shown += 1
                mag = float((e[t] - good[t]).abs().max())
                print(f"  rep {rep} tile {t} (t%4={t % 4}, batch%4={(t // 4) % 4}): {int(diff.sum())} wrong entries, rows {rows}, "
                      f"tuples mb={mbs} q={qs} r={rs}, max |diff| {mag:.3g}, equals tile{stale:+d}" if stale is not None else
                      f"  rep {rep} tile {t} (t%4={t % 4}, batch%4={(t // 4) % 4}): {int(diff.sum())} wrong entries, rows {rows}, "
                      f"tuples mb={mbs} q={qs} r={rs}, max |diff| {mag:.3g}", flush=True)
                if shown <= 3:
                    mb = mbs[0]
                    print("     wrong:", [round(float(x), 4) for x in e[t][rows[0], 16 * mb:16 * mb + 16].tolist()])
                    print("     right:", [round(float(x), 4) for x in good[t][rows[0], 16 * mb:16 * mb + 16].tolist()])
        del e
    print(f"variant {which}: signature over {reps} repetitions:", stats, flush=True)


if __name__ == "__main__" and sys.argv[1] == "sig":
    signature(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)


def featsig(which, reps):
    """For tiles whose e1 differs between repetitions: do the stored layer-1 features (written from the registers the
    LDS rows are written from) differ too, and in which columns?"""
    import torch
    from aether_amd import _lib
    if which != "ref":
        _lib.LIB_PATH = os.path.join(REPO, "aether_amd", f"libaether_hip_haz{which}.so")
    from aether_amd.nn.state2state.aether import Aether
    from aether_amd.synthetic import make_batch
    torch.manual_seed(1)
    m = Aether(4, 64, 0.0, 2, device="cuda")
    m.flags = _lib.FLAG_FORCE_STREAMED | _lib.FLAG_KEEP_INTERMEDIATES
    B, N = 32, 1024
    inp = make_batch(B, N, 2, seed=3, device="cuda")
    Nn, E = B * N, inp["edges"][0].numel()

    def fwd():
        with torch.no_grad():
            m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
        return m.debug_fetch("e1", Nn, E, 64).view(-1, 16, 64), m.debug_fetch("efeat", Nn, E, 32).view(-1, 16, 32)

    e0, f0 = fwd()
    e0, f0 = e0.clone(), f0.clone()
    cols = {}
    n_e = n_f = 0
    shown = 0
    for rep in range(reps):
        e, f = fwd()
        bad = (e != e0).flatten(1).any(1).nonzero().flatten().tolist()
        for t in bad:
            n_e += 1
            df = (f[t] != f0[t])
            if bool(df.any()):
                n_f += 1
                key = tuple(df.any(0).nonzero().flatten().tolist())
                cols[key] = cols.get(key, 0) + 1
                if shown < 4:
                    shown += 1
                    r = df.any(1).nonzero().flatten().tolist()
                    print(f"  rep {rep} tile {t} t%4={t % 4}: feature columns {list(key)} differ in rows {r}")
                    print("     this run :", [round(float(x), 5) for x in f[t][r[0]][:18].tolist()])
                    print("     first run:", [round(float(x), 5) for x in f0[t][r[0]][:18].tolist()])
        del e, f
    print(f"variant {which}: {n_e} tile differences in e1 over {reps} repetitions, {n_f} of them with different stored "
          f"features; differing feature columns: {cols}", flush=True)


if __name__ == "__main__" and sys.argv[1] == "featsig":
    featsig(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)


def forensic(which, reps):
    """Decompose the wrong value: feature 0 = R_i[0][0] rel_0 + R_i[1][0] rel_1 (rel = p_j - p_i).  Which term is off?"""
    import torch
    from aether_amd import _lib
    if which != "ref":
        _lib.LIB_PATH = os.path.join(REPO, "aether_amd", f"libaether_hip_haz{which}.so")
    from aether_amd.nn.state2state.aether import Aether
    from aether_amd.synthetic import make_batch
    torch.manual_seed(1)
    m = Aether(4, 64, 0.0, 2, device="cuda")
    m.flags = _lib.FLAG_FORCE_STREAMED | _lib.FLAG_KEEP_INTERMEDIATES
    B, N = 32, 1024
    inp = make_batch(B, N, 2, seed=3, device="cuda")
    Nn, E = B * N, inp["edges"][0].numel()
    perm = m.graph_perm(inp["edges"], Nn)
    send, recv = inp["edges"][0][perm], inp["edges"][1][perm]

    def fwd():
        with torch.no_grad():
            m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
        return m.debug_fetch("efeat", Nn, E, 32)

    f0 = fwd().clone()
    R = m.debug_fetch("R", Nn, E, 4).clone()          # [R00 R01 R10 R11]
    fld = m.debug_fetch("field", Nn, E, 2).clone()
    can = m.debug_fetch("canon", Nn, E, 4).clone()    # [cv0 cv1 cf0 cf1]
    x, v = inp["x"], inp["vel"]
    shown = 0
    for rep in range(reps):
        f = fwd()
        bad = (f[:, 0] != f0[:, 0]).nonzero().flatten()
        if bad.numel() == 0:
            continue
        for k in bad[:: max(1, bad.numel() // 6)][:6].tolist():
            j, i = int(send[k]), int(recv[k])
            rel = (x[j] - x[i]).double()
            Ri = R[i].double()
            t1, t2 = float(Ri[0] * rel[0]), float(Ri[2] * rel[1])
            a, b = float(f[k, 0]), float(f0[k, 0])
            right = t1 + t2
            wrong, good = (a, b) if abs(b - right) < abs(a - right) else (b, a)
            cands = {"R00": Ri[0], "R01": Ri[1], "R10": Ri[2], "R11": Ri[3], "f_i0": fld[i][0], "f_i1": fld[i][1],
                     "cv_i0": can[i][0], "cv_i1": can[i][1], "cf_i0": can[i][2], "cf_i1": can[i][3],
                     "p_i0": x[i][0], "p_i1": x[i][1], "v_i0": v[i][0], "v_i1": v[i][1],
                     "p_j0": x[j][0], "p_j1": x[j][1], "rel0": rel[0], "rel1": rel[1], "zero": 0.0}
            # wrong = A * rel0 + t2  or  t1 + A * rel1: which known quantity is A?
            A0 = (wrong - t2) / float(rel[0]); A1 = (wrong - t1) / float(rel[1])
            n0 = min(cands, key=lambda c: abs(float(cands[c]) - A0)); n1 = min(cands, key=lambda c: abs(float(cands[c]) - A1))
            print(f"  edge {k} (lane {k % 64}) j={j} i={i}: right {good:.5f} (= {right:.5f}), wrong {wrong:.5f}; "
                  f"R00*rel0={t1:.5f} R10*rel1={t2:.5f}; if the first factor were A: A={A0:.5f} (nearest {n0}={float(cands[n0]):.5f}); "
                  f"if the second: A={A1:.5f} (nearest {n1}={float(cands[n1]):.5f})", flush=True)
            shown += 1
        if shown >= 18:
            break


if __name__ == "__main__" and sys.argv[1] == "forensic":
    forensic(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)
