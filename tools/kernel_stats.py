#!/usr/bin/env python3
"""Print per-kernel register / LDS / instruction-mix stats from a hipcc -save-temps .s file."""
import os, re, subprocess, sys, tempfile
if len(sys.argv) > 1 and sys.argv[1].endswith(".s"):
    txt = open(sys.argv[1]).read()
else:   # compile the library's device code with -save-temps into a scratch dir
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(repo, "aether_amd", "csrc", "aether_hip.hip")
    tmp = tempfile.mkdtemp(prefix="aether_isa_")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
                    "-save-temps", src, "-o", "dev.o"], cwd=tmp, check=True,
                   stderr=subprocess.DEVNULL)
    sfile = [f for f in os.listdir(tmp) if f.endswith(".s")][0]
    txt = open(os.path.join(tmp, sfile)).read()
    print("# ISA in", os.path.join(tmp, sfile))
# metadata
meta = {}
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
    meta[name] = dict(agpr=int(blk.split()[0]), vgpr=g("vgpr_count"), sgpr=g("sgpr_count"),
                      spill=g("vgpr_spill_count"), lds=g("group_segment_fixed_size"),
                      scratch=g("private_segment_fixed_size"))
# bodies
for name, m in meta.items():
    if "hipcub" in name or "rocprim" in name:
        continue
    start = txt.find("\n" + name + ":")
    end = txt.find("s_endpgm", start)
    body = txt[start:end] if start >= 0 else ""
    cnt = lambda pat: len(re.findall(pat, body))
    print(f"{name[:70]:70s} vgpr={m['vgpr']:3d} agpr={m['agpr']:3d} sgpr={m['sgpr']:3d} spill={m['spill']} scratch={m['scratch']} "
          f"mfma={cnt(r'v_mfma')} ds_r128={cnt(r'ds_read_b128')} ds_r32={cnt(r'ds_read_b32')} ds_w={cnt(r'ds_write')} "
          f"flat={cnt(r'flat_load')} gld={cnt(r'global_load')} gst={cnt(r'global_store')} exp={cnt(r'v_exp_f32')} "
          f"rcp={cnt(r'v_rcp_f32')} div_fmas={cnt(r'v_div_fmas')} lines={body.count(chr(10))}")
