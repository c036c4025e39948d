"""Reader for the section dumps written by oracle/_ref/ref_host (oracle/ref_driver.cpp)."""
import struct

import numpy as np


def read_dump(path: str) -> dict:
    out = {}
    with open(path, "rb") as f:
        blob = f.read()
    assert blob[:4] == b"SHRD", "not a ref_host dump"
    pos = 4
    while pos < len(blob):
        (name_len,) = struct.unpack_from("<I", blob, pos)
        pos += 4
        name = blob[pos:pos + name_len].decode()
        pos += name_len
        (count,) = struct.unpack_from("<Q", blob, pos)
        pos += 8
        out[name] = np.frombuffer(blob, dtype="<f4", count=count, offset=pos).copy()
        pos += 4 * count
    return out
