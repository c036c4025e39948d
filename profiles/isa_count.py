#!/usr/bin/env python3
"""Static instruction counts per kernel of a built library, by class (VALU, SALU, VMEM, LDS, ...), from its
disassembly:   python profiles/isa_count.py [shader-ray_amd/libshray_hip.so] [substring of the kernel name]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def classify(op):
    if op.startswith(("v_", "v_pk")):
        return "valu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setprio", "s_sleep")):
        return "control"
    if op.startswith("s_"):
        return "salu"
    return "other"


def disassemble(lib):
    """{mangled kernel name: [opcodes]} of the gfx950 code objects inside a shared library / object file (one offload
    bundle per translation unit, back to back in the .hip_fatbin section) or of a plain code object."""
    texts = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat],
                           capture_output=True, text=True)
        blob = open(fat, "rb").read() if r.returncode == 0 and os.path.exists(fat) else b""
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        objects = []
        for k, at in enumerate(starts):
            piece = os.path.join(tmp, f"bundle{k}.bin")
            open(piece, "wb").write(blob[at:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f"gfx950_{k}.co")
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={piece}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            objects.append(co)
        if not objects:
            objects = [lib]    # already a code object
        for co in objects:
            texts.append(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co],
                                        capture_output=True, text=True).stdout)
    kernels, cur = {}, None
    for line in "\n".join(texts).splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line + " ")
        if m and cur is not None:
            cur.append(m.group(1))
    return kernels


def counts(ops):
    c = collections.Counter(classify(op) for op in ops)
    c["total"] = len(ops)
    return dict(c)


def demangle(names):
    try:
        return subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    except FileNotFoundError:
        return names


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "shader-ray_amd", "libshray_hip.so")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    ks = disassemble(lib)
    names = list(ks)
    for name, pretty in zip(names, demangle(names)):
        pretty = re.sub(r"\(.*", "", pretty.replace("shray::", "").replace("void ", ""))
        if want in pretty and not name.endswith(".kd"):
            c = counts(ks[name])
            print(f"{pretty:<60} " + " ".join(f"{k} {c.get(k, 0):>5}" for k in ("total", "valu", "salu", "vmem", "lds", "smem", "control")))
