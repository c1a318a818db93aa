"""Register / spill / LDS figures of every kernel in libgnxr.so, read from the code-object notes (dev tool + CPU test helper).

usage: python tools/kernel_regs.py [lib.so] [--all]     prints one line per kernel, worst first

The .hip_fatbin section of the library is a sequence of clang offload bundles (one per HIP translation unit); each bundle's gfx950 entry
is an ELF code object whose NT_AMDGPU_METADATA note carries, per kernel, .vgpr_count / .agpr_count / .sgpr_spill_count / .vgpr_spill_count /
.private_segment_fixed_size / .group_segment_fixed_size.  On gfx950 VGPRs and AGPRs share one file of 512 per SIMD lane, so
waves per SIMD = floor(512 / align8(.vgpr_count)) (.vgpr_count already includes the AGPRs): 257 registers run ONE wave per SIMD, 256 run two.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """yields the gfx950 ELF images embedded in `lib`"""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        data = open(fat, "rb").read()
    pos = data.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size > 0:
                yield data[pos + off:pos + off + size]
        pos = data.find(MAGIC, pos + len(MAGIC))


def kernels(lib=None):
    """[{name, vgpr, agpr, sgpr, sgpr_spill, vgpr_spill, scratch, lds, waves_per_simd}] for every kernel of the library"""
    lib = lib or os.path.join(ROOT, "gnxraytracer_amd", "libgnxr.so")
    out = []
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk

            def num(key, blk=blk):
                m = re.search(r"\." + key + r":\s+(\d+)", blk)
                return int(m.group(1)) if m else 0

            m = re.search(r"\.name:\s+(\S+)", blk)
            if not m:
                continue
            name = m.group(1)
            try:
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
            regs = num("vgpr_count")   # on gfx90a+ .vgpr_count is the unified total (architectural VGPRs + AGPRs)
            out.append(dict(name=name, vgpr=num("vgpr_count"), agpr=num("agpr_count"), sgpr=num("sgpr_count"), sgpr_spill=num("sgpr_spill_count"),
                            vgpr_spill=num("vgpr_spill_count"), scratch=num("private_segment_fixed_size"), lds=num("group_segment_fixed_size"),
                            waves_per_simd=max(1, min(8, 512 // max(8, (regs + 7) // 8 * 8)))))
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ks = kernels(args[0] if args else None)
    ks.sort(key=lambda k: (-k["vgpr"], -k["sgpr_spill"]))
    for k in ks if "--all" in sys.argv else ks[:40]:
        print(f"{k['vgpr']:4d}v {k['agpr']:3d}a {k['sgpr']:4d}s  spill s{k['sgpr_spill']:4d} v{k['vgpr_spill']:4d}  scratch {k['scratch']:5d}  lds {k['lds']:6d}  waves {k['waves_per_simd']}  {k['name'][:150]}")
