"""cv::imread replacement (fh_imread / fh_image_decode, csrc/image_io.cpp) against the committed fixtures:
files written by tests/golden/make_images.py with the pixels Pillow (libjpeg-turbo / zlib, the codecs behind
OpenCV's imread) decodes from them.  Bit-exact: JPEG decode is integer arithmetic end to end."""
import ctypes as C
import os

import numpy as np
import pytest

import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
IMAGES = os.path.join(HERE, "golden", "images")
EXPECTED = np.load(os.path.join(HERE, "golden", "images_expected.npz"))


def decode(data: bytes):
    L = fa.lib()
    p, r, c = C.c_void_p(), C.c_int(), C.c_int()
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    rc = L.fh_image_decode(C.cast(buf, C.c_void_p), len(data), C.byref(p), C.byref(r), C.byref(c))
    if rc != 0:
        return None
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), (r.value, c.value, 3)).copy()
    finally:
        L.fh_image_free(p)


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
def test_fixture_decodes_bit_exact(name):
    with open(os.path.join(IMAGES, name), "rb") as f:
        got = decode(f.read())
    assert got is not None, _lib.last_error()
    exp = EXPECTED[name]
    assert got.shape == exp.shape
    if not np.array_equal(got, exp):
        d = np.abs(got.astype(int) - exp.astype(int))
        pytest.fail(f"{name}: {np.count_nonzero(d)} of {d.size} samples differ, max {d.max()}, first at {np.argwhere(d)[0]}")


def test_c1_jpeg_640_sha256():
    import hashlib, json
    with open(os.path.join(HERE, "golden", "images_sha256.json")) as f:
        sha = json.load(f)
    img = fa.imread(os.path.join(IMAGES, "c1_640x640.jpg"))
    assert img is not None and img.shape == (640, 640, 3)
    assert hashlib.sha256(img.tobytes()).hexdigest() == sha["c1_640x640.jpg"]


def test_imread_path_and_api_wrapper(tmp_path):
    name = "j420_q75_odd.jpg"
    img = fa.imread(os.path.join(IMAGES, name))
    assert img.dtype == np.uint8 and np.array_equal(img, EXPECTED[name])
    assert fa.imread(str(tmp_path / "missing.jpg")) is None and "cannot open" in _lib.last_error()   # cv::imread: empty Mat


def test_corrupt_and_unsupported_inputs_fail_cleanly():
    with open(os.path.join(IMAGES, "j420_q75_odd.jpg"), "rb") as f:
        data = f.read()
    assert decode(b"") is None
    assert decode(b"not an image at all") is None and "unrecognised" in _lib.last_error()
    assert decode(data[:20]) is None                                       # header cut short
    cut = decode(data[: len(data) // 2])                                   # entropy data cut: libjpeg-style, rest decodes as zeros
    assert cut is not None and cut.shape == EXPECTED["j420_q75_odd.jpg"].shape
    with open(os.path.join(IMAGES, "p_rgb.png"), "rb") as f:
        png = f.read()
    assert decode(png[:60]) is None and "PNG" in _lib.last_error()
    huge = bytearray(data)                                                  # a header that claims 65535 x 65535 pixels
    sof = data.index(b"\xff\xc0")
    huge[sof + 5: sof + 9] = b"\xff\xff\xff\xff"
    assert decode(bytes(huge)) is None and "unreasonable" in _lib.last_error()
    rng = np.random.default_rng(0)
    for _ in range(200):                                                   # bit flips must never crash the decoder
        b = bytearray(data)
        for i in rng.integers(2, len(b), 3):
            b[i] ^= 1 << int(rng.integers(0, 8))
        decode(bytes(b))


def test_random_images_against_pillow_live():
    """Wider sweep when Pillow is importable (it is in this image): sizes, qualities, sampling modes."""
    Image = pytest.importorskip("PIL.Image")
    import io
    rng = np.random.default_rng(7)
    for i in range(60):
        h, w = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if i % 2:
            arr = (arr // 8 + np.linspace(0, 200, w, dtype=np.uint8)[None, :, None]).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(arr, "RGB").save(buf, "JPEG", quality=int(rng.integers(5, 101)), subsampling=int(rng.integers(0, 3)),
                                         progressive=bool(i % 3 == 0), optimize=bool(i % 5 == 0))
        exp = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))[:, :, ::-1]
        got = decode(buf.getvalue())
        assert got is not None, _lib.last_error()
        assert np.array_equal(got, exp), (i, h, w)
