"""Background / environment loading of the host layer (host/background.cpp): the reference's
argument forms (ray.cpp:1002-1035) and the Radiance RGBE reader that stands in for its
FreeImagePlus file branch (ray.cpp:1036-1074)."""
import numpy as np
import pytest


def float_to_rgbe(rgb):
    """Greg Ward's float -> RGBE (the encoder every .hdr writer uses)."""
    rgb = np.asarray(rgb, dtype=np.float64)
    v = rgb.max(axis=-1)
    out = np.zeros(rgb.shape[:-1] + (4,), dtype=np.uint8)
    ok = v > 1e-32
    m, e = np.frexp(v[ok])
    scale = m * 256.0 / v[ok]
    out[ok, :3] = np.clip(rgb[ok] * scale[:, None], 0, 255).astype(np.uint8)
    out[ok, 3] = (e + 128).astype(np.uint8)
    return out


def rle_channel(values):
    """New-style Radiance RLE of one channel of a scanline."""
    out = bytearray()
    i, n = 0, len(values)
    while i < n:
        run = 1
        while i + run < n and run < 127 and values[i + run] == values[i]:
            run += 1
        if run >= 4:
            out += bytes([128 + run, values[i]])
            i += run
        else:
            j = i
            while j < n and j - i < 128:
                r = 1
                while j + r < n and r < 4 and values[j + r] == values[j]:
                    r += 1
                if r >= 4:
                    break
                j += 1
            out += bytes([j - i]) + bytes(values[i:j])
            i = j
    return bytes(out)


def write_hdr(path, rgbe_top_down, rle):
    h, w, _ = rgbe_top_down.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\n# synthetic test picture\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n")
        f.write(b"-Y %d +X %d\n" % (h, w))
        for row in rgbe_top_down:
            if rle:
                f.write(bytes([2, 2, w >> 8, w & 255]))
                for ch in range(4):
                    f.write(rle_channel(row[:, ch].tolist()))
            else:
                f.write(row.tobytes())


def expected_floats(rgbe_top_down):
    e = rgbe_top_down[..., 3].astype(np.int32)
    f = np.where(e == 0, 0.0, np.ldexp(1.0, e - 136))
    img = rgbe_top_down[..., :3].astype(np.float64) * f[..., None]
    return img[::-1].astype(np.float32)          # row 0 = bottom row


@pytest.mark.parametrize("rle", [False, True])
def test_radiance_hdr_reader(pkg, tmp_path, rle):
    rng = np.random.default_rng(11)
    w, h = 40, 9
    img = rng.uniform(0, 2, (h, w, 3))
    img[2, 5:30] = [0.25, 0.5, 60.0]      # a long run (exercises RLE runs) with an HDR value
    img[4] = 0.0                           # black row: exponent 0
    rgbe = float_to_rgbe(img)
    path = str(tmp_path / ("rle.hdr" if rle else "flat.hdr"))
    write_hdr(path, rgbe, rle)
    got = pkg.load_background(path)
    assert got.shape == (h, w, 3) and got.dtype == np.float32
    assert np.array_equal(got, expected_floats(rgbe))
    assert got.max() > 50 and np.all(got[h - 1 - 4] == 0)


def test_background_argument_forms(pkg):
    c = pkg.load_background("0.25, 0.5, 2")
    assert c.shape == (1, 1, 3) and c.reshape(-1).tolist() == [0.25, 0.5, 2.0]
    hx = pkg.load_background("ff8000")
    assert np.allclose(hx.reshape(-1), [1.0, 128 / 255.0, 0.0])
    grid = pkg.load_background("grid")
    assert np.array_equal(grid, pkg.scenes.environment_grid(2048))   # the generator used by the tests matches it


def test_background_errors(pkg, tmp_path):
    with pytest.raises(RuntimeError):
        pkg.load_background(str(tmp_path / "missing.hdr"))
    bad = tmp_path / "bad.hdr"
    bad.write_bytes(b"P6 2 2 255\n" + bytes(12))
    with pytest.raises(RuntimeError):
        pkg.load_background(str(bad))
    cut = tmp_path / "cut.hdr"
    cut.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 16\n" + bytes(40))
    with pytest.raises(RuntimeError):
        pkg.load_background(str(cut))
