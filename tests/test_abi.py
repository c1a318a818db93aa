"""C-ABI library: loads, exports every symbol include/gnxr.h declares, struct layouts match (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_header_symbols_exported(gx):
    hdr = open(os.path.join(ROOT, "include", "gnxr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gnxr_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    lib = C.CDLL(gx.LIB_PATH)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, f"libgnxr.so does not export: {missing}"
    # and the Python binding table covers the same set
    from gnxraytracer_amd import _abi
    assert sorted(_abi.PROTOTYPES) == declared


def test_struct_sizes(gx):
    from gnxraytracer_amd import _abi
    for i, st in enumerate(_abi.ABI_STRUCTS):
        assert gx.lib().gnxr_abi_sizeof(i) == C.sizeof(st), st.__name__
    assert gx.lib().gnxr_abi_version() == _abi.GNXR_ABI_VERSION


def test_no_cpu_fallback(gx):
    """Without a HIP device every compute entry point must fail loudly with GNXR_ERR_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the failure path cannot be observed")
    import scenes
    b = scenes.cornell()
    with pytest.raises(gx.GnxrError) as e:
        gx.Scene(b)
    assert "no HIP device" in str(e.value) or "-2" in str(e.value)
    with pytest.raises(gx.GnxrError):
        gx.sample_halton(64, 64, [0], [0], [0], [0])


def test_product_does_not_reference_oracle():
    """The oracle is test infrastructure: nothing in the package may import, include, link or dlopen it."""
    pkg = os.path.join(ROOT, "gnxraytracer_amd")
    bad = re.compile(r"^\s*(import|from)\s+\S*oracle|#\s*include\s*[\"<][^\">]*oracle|libgnx_oracle|gnxo_|dlopen\([^)]*oracle|CDLL\([^)]*oracle")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".cpp", ".hip")):
                for line in open(os.path.join(dp, f), errors="replace"):
                    assert not bad.search(line), (f, line)


def test_c_caller_builds_and_fails_loudly_without_a_device(gx):
    """tools/gnxr_cli.c (RenderThread::run against the C ABI) compiles as C11 against include/gnxr.h and links libgnxr.so; without a
    HIP device it reports GNXR_ERR_NO_DEVICE and exits with status 3 instead of rendering on the CPU."""
    import subprocess, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as ge
    cli = ge.build_cli()
    assert subprocess.run([cli, "--help"], capture_output=True).returncode == 0
    assert subprocess.run([cli], capture_output=True).returncode == 2          # --out is required
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the failure path cannot be observed")
    r = subprocess.run([cli, "--out", "/tmp/gnxr_cli_should_not_exist.png"], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr and not os.path.exists("/tmp/gnxr_cli_should_not_exist.png")


def test_kernel_register_budgets(gx):
    """Reads the code-object notes of every kernel in libgnxr.so (tools/kernel_regs.py): no kernel may need more than 256 registers
    (VGPRs + AGPRs: 257 halve the occupancy to one wave per SIMD -- round 2 shipped the Disney-class k_shade at exactly 257), the
    traversal kernel keeps its five waves per SIMD without spilling, and the shade kernels of the headline path hold the three waves per
    SIMD they were measured to want (the glossy class pays for them with ~100 spilled dwords of scratch: bounded here)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_regs
    ks = kernel_regs.kernels(gx.LIB_PATH)
    assert len(ks) > 100, len(ks)
    over = [(k["name"], k["vgpr"]) for k in ks if k["vgpr"] > 256]
    assert not over, over
    t4 = [k for k in ks if "k_trace4<false, false" in k["name"]]
    assert t4 and all(k["waves_per_simd"] >= 5 and k["vgpr_spill"] == 0 for k in t4), t4
    head = [k for k in ks if ("k_shade<458879u, 1, false, false>" in k["name"] or "k_shade<3u, 1, false, false>" in k["name"])]
    assert len(head) == 2 and all(k["scratch"] <= 256 and k["waves_per_simd"] >= 3 for k in head), head


def test_library_was_built_from_the_sources_in_the_tree():
    """build() writes the hash of csrc/ + flags it compiled beside libgnxr.so; a library left behind by an experiment build or by a checkout of
    other sources (what ships to the GPU box is the file in the tree, not the sources) must not pass for the product."""
    import __graft_entry__ as ge
    ge.build_lib()
    assert ge.library_source_id() == ge.source_id()
