#!/usr/bin/env python3
"""ISA invariants of libaether_hip.so, checked on the built code object.

Scans the gfx950 disassembly of every kernel.

  R3  (FAILS the check) a packed fp32 VALU instruction -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- with an op_sel
      bit on src0 or src1, i.e. whose LOW result lane reads the HIGH half of a multiplicand / addend.  Measured on the
      MI355X (tools/micro/pkfma_mfma.hip): while the SIMD's other wave issues bf16 MFMAs this form returns a wrong low
      result in lanes 48-63 (7e-3 of lane-results next to v_mfma_f32_16x16x32_bf16); op_sel on src2 only, and cleared
      op_sel_hi bits, never failed.  This -- not the accumulator patterns below -- is what corrupted ~3 tiles per
      million in the uncapped k_edge_layer1 of round 2 (tools/hazard_variants.py: variants 0-9).  Any kernel may share
      a SIMD with another kernel's MFMA waves, so the rule covers the whole library.

  R1  (reported with --all) a memory load (ds_read* / global_load* / ...) whose destination is an accumulator register
      a[..] that a later MFMA reads as SrcC;
  R2  (reported with --all) an MFMA whose destination tuple partially overlaps its SrcC tuple.
      Round 2 suspected R1; isolated (tools/micro/agpr_hazard.hip: 8e8 trials each of R1, R2 and the compiler's whole
      sliding-tuple sequence, bit-exact) and in the kernel (2 x 2 variants: every combination failed alike until the
      packed FMAs were removed, then none did) both patterns are sound, so they do not fail the check.

  R4  (FAILS the check, library input only) a kernel that takes a by-value struct of pointer arrays (FilterTypes: the
      k_s2s_filter_*_types / k_dyn_filter_combine family, nodes of every captured variable-N step) must consume NO implicit
      ("hidden") kernel argument.  Round 3's k_s2s_filter_split_types<15> read gridDim.x (hidden_block_count_x at the end
      of a 480-byte kernarg segment) on a 2-D grid; as a graph node replayed back to back it ended in a GPU memory access
      fault (20 of 20 runs).  With the workgroup count as an explicit argument the same reproducer passes.

  R5  (FAILS the check) an instruction that names the destination of a `ds_read_b64_tr_b16` still in flight (issued from inline
      asm, waited for by a hand-placed s_waitcnt).  Round 4's gemm_split_T issued two reads per asm statement with plain "=v"
      outputs: in k_fused_bwd<3> the compiler gave the first read's destination the register of the address (its last use),
      so the second read took its address from a register the first read was about to overwrite -- right whenever the second
      read issued before the first one's data landed, i.e. nearly always: one test failure in three full runs.  The outputs
      are early-clobber now; this rule keeps it that way.

  R6  (FAILS the check, library input only) register placement the compiler chose behind the source's back, found twice in
      round 4 by their cost, not by a wrong result:
      (a) the inference instances of k_fused (KEEP = false; one or two tiles per wave -- the three-tile instance spills by
          design, HISTORY 4.1b) must not touch scratch memory (private segment 0 bytes): two
          floats assigned through a by-reference lambda capture were kept in scratch -- 8 bytes per thread stored and
          re-loaded in the feature phase, + 1 MB of WRITE_SIZE per launch;
      (b) k_s2s_gemm_split / k_s2s_gemm_split_r1 / k_wgemm must contain no v_accvgpr_read / v_accvgpr_write: built for
          one wave per SIMD (__launch_bounds__(256, 1)) the compiler selects the AGPR form for every MFMA and copied all 64
          accumulators of a lane out and back around the rarely taken rescale branch in EVERY k step (+ 0.4 ms per seq2seq
          step at N = 20).

Usage: isa_check.py [libaether_hip.so | file.s] [--all] [--kernel SUBSTR]
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(os.path.dirname(HERE), "aether_amd", "libaether_hip.so")

_REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+))")
_LOADS = ("ds_read", "ds_load", "global_load", "buffer_load", "scratch_load", "flat_load")


def disassemble(path: str) -> str:
    """gfx950 disassembly of a shared library's embedded code object (or the text of a .s file)."""
    if path.endswith(".s"):
        return open(path).read()
    with tempfile.TemporaryDirectory() as td:
        co, fat = os.path.join(td, "dev.co"), os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", path],
                       check=True, capture_output=True)
        bundler = os.path.join(LLVM, "clang-offload-bundler")
        subprocess.run([bundler, "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
        if not os.path.exists(co) or os.path.getsize(co) == 0:
            raise RuntimeError(f"no gfx950 code object in {path}")
        out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co],
                             check=True, capture_output=True, text=True).stdout
    return out


def kernel_notes(path: str):
    """{kernel name: (kernarg segment bytes, [hidden_* argument kinds])} from the code object's metadata notes."""
    with tempfile.TemporaryDirectory() as td:
        co, fat = os.path.join(td, "dev.co"), os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", path],
                       check=True, capture_output=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"],
                       check=True, capture_output=True)
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                             text=True).stdout
    out = {}
    for blk in txt.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        seg = re.search(r"\.kernarg_segment_size:\s+(\d+)", blk)
        priv = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        if name:
            out[name.group(1)] = (int(seg.group(1)) if seg else -1, re.findall(r"\.value_kind:\s+(hidden_\w+)", blk),
                                  int(priv.group(1)) if priv else 0)
    return out


R4_KERNELS = ("k_s2s_filter_split_types", "k_s2s_filter_bimg_types", "k_dyn_filter_combine", "k_dynb_", "k_wgemm", "k_wide_prep")


def check_hidden_args(path: str):
    """Rule R4 -> [(kernel, hidden kinds)] of the by-value-struct kernels that consume hidden kernel arguments."""
    return [(n, h) for n, (_, h, _p) in kernel_notes(path).items() if h and any(k in n for k in R4_KERNELS)]


R6_NO_AGPR_COPIES = ("k_s2s_gemm_split", "k_wgemm")


def check_scratch(path: str):
    """Rule R6a -> [(kernel, private segment bytes)] of inference k_fused instances that use scratch memory."""
    return [(n, p) for n, (_, _h, p) in kernel_notes(path).items() if p and re.search(r"7k_fusedILi\dELi\d+ELi[12]ELb0E", n)]


def split_kernels(txt: str):
    """Yield (name, [instruction text]) for every function of a disassembly or compiler .s file."""
    cur, body = None, []
    for ln in txt.split("\n"):
        m = re.match(r"^[0-9a-f]* ?<([^>]+)>:$", ln) or re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith((".", "\t", " ")):
            if cur is not None:
                yield cur, body
            cur, body = m.group(1), []
            continue
        t = ln.strip()
        if cur is None or not t or t.startswith((".", ";", "//")):
            continue
        t = t.split("//")[0].split(";")[0].strip()
        if t and not t.endswith(":"):
            body.append(t)
    if cur is not None:
        yield cur, body


def regs(tok: str):
    """('a'|'v', first, last) of a register operand, or None."""
    m = _REG.search(tok)
    if not m:
        return None
    if m.group(2) is not None:
        return m.group(1), int(m.group(2)), int(m.group(3))
    return m.group(1), int(m.group(4)), int(m.group(4))


def operands(ins: str):
    parts = ins.split(None, 1)
    return [p.strip() for p in parts[1].split(",")] if len(parts) > 1 else []


_PK = re.compile(r"^v_pk_(fma|mul|add|max|min)\w*_f32\b")
_OPSEL = re.compile(r"op_sel:\[([01,]+)\]")


def check_kernel(body):
    """-> dict(bf16, r1 = [(load, mfma)], r2 = [mfma], r3 = [packed instruction], agpr_loads)"""
    pending = {}                 # agpr index -> load instruction that last wrote it, nothing has overwritten it since
    r1, r2, r3 = [], [], []
    bf16 = False
    n_agpr_loads = 0
    for ins in body:
        op = ins.split()[0]
        ops = operands(ins)
        if _PK.match(op):
            m = _OPSEL.search(ins)
            if m and "1" in m.group(1).split(",")[:2]:
                r3.append(ins)
            continue
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            if "bf16" in op:
                bf16 = True
            d, c = regs(ops[0]), regs(ops[3]) if len(ops) > 3 else None
            if c and c[0] == "a":
                hit = sorted({pending[r] for r in range(c[1], c[2] + 1) if r in pending})
                for ld in hit:
                    r1.append((ld, ins))
            if d and c and d[0] == c[0] and (d[1], d[2]) != (c[1], c[2]) and not (d[2] < c[1] or c[2] < d[1]):
                r2.append(ins)
            if d and d[0] == "a":
                for r in range(d[1], d[2] + 1):
                    pending.pop(r, None)
            continue
        if op.startswith(_LOADS) and ops:
            d = regs(ops[0])
            if d and d[0] == "a" and ops[0].lstrip().startswith("a"):
                n_agpr_loads += 1
                for r in range(d[1], d[2] + 1):
                    pending[r] = ins
            continue
        if op.startswith("v_accvgpr_write") and ops:
            d = regs(ops[0])
            if d:
                pending.pop(d[1], None)
    return dict(bf16=bf16, r1=r1, r2=r2, r3=r3, agpr_loads=n_agpr_loads)


_LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def check_async_lds_dest(body):
    """Rule R5 -> [(instruction, (first, last))]: an instruction that names a register a `ds_read_b64_tr_b16` (issued from
    inline asm, waited for by a hand-written `s_waitcnt lgkmcnt(n)`) has not yet delivered.  The compiler believes an asm
    statement's outputs are ready when the statement ends: it may hand the same register to a later operand of that statement
    (outputs that are not early-clobber: the address of the statement's second read) or copy it before the wait.  Either
    reads or overwrites a register with a load still in flight -- works until the load happens to land first.
    Linear scan; LDS and scalar-memory operations are taken to complete in order (true for LDS alone)."""
    outstanding, viol = [], []
    for ins in body:
        op = ins.split()[0]
        if op == "s_waitcnt":
            m = _LGKM.search(ins)
            if m:
                n = int(m.group(1))
                outstanding = outstanding[len(outstanding) - n:] if n else []
            continue
        live = [d for d in outstanding if d]
        if live:
            for tok in operands(ins):
                r = regs(tok)
                hit = next((d for d in live if r and r[0] == "v" and not (r[2] < d[0] or d[1] < r[1])), None)
                if hit:
                    viol.append((ins, hit))
                    break
        if op.startswith(("ds_", "s_load", "s_buffer_load")):
            d = None
            if op.startswith("ds_read_b64_tr"):
                r = regs(operands(ins)[0])
                d = (r[1], r[2])
            outstanding.append(d)
    return viol


def main(argv):
    path = next((a for a in argv if not a.startswith("--")), DEFAULT_LIB)
    show_all = "--all" in argv
    only = argv[argv.index("--kernel") + 1] if "--kernel" in argv else None
    txt = disassemble(path)
    bad = 0
    n = 0
    for name, body in split_kernels(txt):
        if only and only not in name:
            continue
        n += 1
        res = check_kernel(body)
        r5 = check_async_lds_dest(body)
        flagged = bool(res["r3"]) or bool(r5)
        if r5:
            print(f"FAIL {name}: R5(register of a transposed LDS read in flight is named before its wait)={len(r5)}")
            for ins, d in r5[:3]:
                print(f"     R5: {ins}   (v[{d[0]}:{d[1]}] in flight)")
        if flagged or (show_all and (res["r1"] or res["r2"] or res["agpr_loads"])):
            print(f"{'FAIL' if flagged else 'note'} {name}: R3(packed fp32 op_sel on src0/src1)={len(res['r3'])} "
                  f"bf16_mfma={res['bf16']} agpr_loads={res['agpr_loads']} R1(load->SrcC)={len(res['r1'])} "
                  f"R2(partial dst/SrcC overlap)={len(res['r2'])}")
            for pk in res["r3"][:3]:
                print("     R3:", pk)
            if show_all:
                for ld, mf in res["r1"][:2]:
                    print("     R1:", ld, "->", mf)
                for mf in res["r2"][:2]:
                    print("     R2:", mf)
        bad += bool(flagged)
        if any(k in name for k in R6_NO_AGPR_COPIES):
            copies = sum(1 for ins in body if "v_accvgpr_read" in ins or "v_accvgpr_write" in ins)
            if copies:
                print(f"FAIL {name}: R6b(AGPR copies in a kernel that is meant to keep its accumulators in VGPRs)={copies}")
                bad += 1
    if not path.endswith(".s") and not only:
        for name, priv in check_scratch(path):
            print(f"FAIL {name}: R6a(inference k_fused uses {priv} bytes of scratch memory per thread)")
            bad += 1
        for name, hidden in check_hidden_args(path):
            print(f"FAIL {name}: R4(by-value struct kernel consumes hidden kernel arguments)={sorted(set(hidden))}")
            bad += 1
    print(f"isa_check: {n} kernels scanned, {bad} failing")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
