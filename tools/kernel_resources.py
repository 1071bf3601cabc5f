#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in a built library, read from the code objects'
AMDGPU metadata notes (no GPU needed).

usage: kernel_resources.py [library.so] [name substring ...]

The library's .hip_fatbin section is a sequence of clang offload bundles (one per translation
unit); each carries one gfx950 code object whose NT_AMDGPU_METADATA note lists, per kernel,
.vgpr_count / .agpr_count / .sgpr_count / .vgpr_spill_count / .private_segment_fixed_size /
.group_segment_fixed_size.  `resources(path)` returns {demangled-ish name: dict}.
Used by tests/test_host_logic.py::test_hot_kernels_do_not_spill (the spill audit)."""
import os
import struct
import subprocess
import sys
import tempfile

import yaml

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
DEFAULT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       "vrp-gym_amd", "vrpgym_hip", "libvrpgym_hip.so")


def code_objects(path):
    """The gfx950 code objects (bytes) bundled into a shared library or object file."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary",
                        "--only-section=.hip_fatbin", path, fat], check=True)
        blob = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return out
        (count,) = struct.unpack_from("<Q", blob, i + 24)
        off = i + 32
        for _ in range(count):
            o, size, tlen = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tlen].decode()
            off += tlen
            if "gfx950" in triple and size:
                out.append(blob[i + o:i + o + size])
        pos = i + 24


def _demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), text=True,
                       capture_output=True, check=True)
    return r.stdout.split("\n")[:len(names)]


def resources(path=DEFAULT):
    table = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(co)
            f.flush()
            r = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], text=True,
                               capture_output=True, check=True)
        text = r.stdout
        a, b = text.find("---"), text.rfind("...")
        if a < 0:
            continue
        meta = yaml.safe_load(text[a:b if b > a else None])
        kernels = meta.get("amdhsa.kernels", [])
        for k, name in zip(kernels, _demangle([k[".name"] for k in kernels])):
            table[name] = {
                "vgpr": k.get(".vgpr_count", 0), "agpr": k.get(".agpr_count", 0),
                "sgpr": k.get(".sgpr_count", 0), "vgpr_spill": k.get(".vgpr_spill_count", 0),
                "sgpr_spill": k.get(".sgpr_spill_count", 0),
                "scratch": k.get(".private_segment_fixed_size", 0),
                "lds": k.get(".group_segment_fixed_size", 0),
                "max_wg": k.get(".max_flat_workgroup_size", 0),
            }
    return table


def main():
    args = sys.argv[1:]
    path = DEFAULT
    if args and os.path.exists(args[0]):
        path = args.pop(0)
    t = resources(path)
    print(f"{'kernel':84s} vgpr agpr sgpr spill scratch    lds")
    for name in sorted(t):
        if args and not any(a in name for a in args):
            continue
        r = t[name]
        short = name.split("(")[0].replace("void ", "")
        print(f"{short[:84]:84s} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['vgpr_spill']:5d} "
              f"{r['scratch']:7d} {r['lds']:6d}")


if __name__ == "__main__":
    main()
