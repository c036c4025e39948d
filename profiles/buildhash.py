"""Hash of the compiled kernels: bench.py uses it to tell whether a committed rocprofv3 counter file
(profiles/rNN/pmc_*.json) was measured on the kernels it is timing.

The hash is taken over the DEVICE CODE of the library that is loaded -- the .hip_fatbin section of libshray_hip.so
(or of $SHRAY_HIP_LIB) -- not over source files: a comment or a host-side change leaves it alone (round 3 keyed the
counter file by a hash of the sources, comments included, and had to un-edit a comment to keep it valid)."""
import hashlib
import os
import struct

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _section(path, wanted):
    """bytes of ELF64 (little-endian) section `wanted`, or None"""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        return None
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def header(k):
        name, _type, _flags, _addr, offset, size = struct.unpack_from("<IIQQQQ", data, shoff + k * shentsize)
        return name, offset, size
    _, stroff, strsize = header(shstrndx)
    names = data[stroff:stroff + strsize]
    for k in range(shnum):
        name, offset, size = header(k)
        if names[name:names.index(b"\0", name)] == wanted.encode():
            return data[offset:offset + size]
    return None


def kernel_source_hash() -> str:
    """(the name is historical: it is the hash of the code objects now)"""
    lib = os.environ.get("SHRAY_HIP_LIB") or os.path.join(ROOT, "shader-ray_amd", "libshray_hip.so")
    blob = _section(lib, ".hip_fatbin")
    if blob is None:
        raise RuntimeError(f"{lib}: no .hip_fatbin section")
    return hashlib.sha256(blob).hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
