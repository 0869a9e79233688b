#!/usr/bin/env python3
"""Register / scratch budget of every gfx950 kernel in the shipped library (code-object notes), and the list of kernels that sit
at the full 512-register budget WITH spills -- the regime in which hipcc (roc-7.2.0) mis-reloaded a split 64-bit spill in
k_compose_smoother<8, true> (scripts/repro_compose8). Usage: list_kernel_resources.py [lib.so] [--all]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


CMAGIC = b"CCOB"      # compressed bundle (--offload-compress): magic, u16 version, u16 method, then (v2: u32, v3: u64) total size, ...


def _bundle_entries(data, i):
    n = struct.unpack_from("<Q", data, i + 24)[0]
    q = i + 32
    out = []
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, q)
        triple = data[q + 24:q + 24 + tl].decode()
        q += 24 + tl
        if "gfx950" in triple and size:
            out.append(data[i + off:i + off + size])
    return out


def code_objects(path):
    """every gfx950 code object of a fat binary: plain offload bundles are parsed here; zstd-compressed ones (CCOB) are cut out
    and handed to clang-offload-bundler, which unpacks them"""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        out += _bundle_entries(data, i)
        pos = i + 24
    pos = 0
    while True:
        i = data.find(CMAGIC, pos)
        if i < 0:
            break
        ver = struct.unpack_from("<H", data, i + 4)[0]
        total = struct.unpack_from("<Q", data, i + 8)[0] if ver >= 3 else struct.unpack_from("<I", data, i + 8)[0]
        if ver not in (1, 2, 3) or total <= 24 or i + total > len(data):
            pos = i + 4
            continue
        with tempfile.TemporaryDirectory() as td:
            src, dst = os.path.join(td, "in.bundle"), os.path.join(td, "out.co")
            open(src, "wb").write(data[i:i + total])
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={src}",
                                f"--output={dst}", "--unbundle"], capture_output=True, text=True)
            if r.returncode == 0 and os.path.exists(dst) and os.path.getsize(dst):
                out.append(open(dst, "rb").read())
        pos = i + total
    return out


def kernels(blob):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(blob)
        name = f.name
    try:
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", name], capture_output=True, text=True).stdout
    finally:
        os.unlink(name)
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda key: re.search(rf"\.{key}:\s*(\S+)", blk)
        name = g("name").group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        out.append(dict(name=dem.replace("(anonymous namespace)::", "").split("(")[0], agpr=int(blk.split()[0]), vgpr=int(g("vgpr_count").group(1)), vspill=int(g("vgpr_spill_count").group(1)),
                        sspill=int(g("sgpr_spill_count").group(1)), scratch=int(g("private_segment_fixed_size").group(1))))
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "temporalgps.jl_amd", "libtgp_hip.so")
    ks = [k for blob in code_objects(lib) for k in kernels(blob)]
    risky = [k for k in ks if k["vgpr"] >= 512 and (k["vspill"] or k["sspill"])]
    print(f"{lib}: {len(ks)} kernels; {sum(1 for k in ks if k['scratch'])} use scratch; {len(risky)} at 512 registers with spills")
    for k in sorted(ks if "--all" in sys.argv else risky, key=lambda k: k["name"]):
        print(f"  {k['name']:90s} vgpr {k['vgpr']:3d} (agpr {k['agpr']:3d}) vgpr_spill {k['vspill']:4d} sgpr_spill {k['sspill']:4d} scratch {k['scratch']:6d} B")
