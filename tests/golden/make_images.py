"""Writes the image-decode fixtures: small JPEG / PNG / BMP / PNM files plus the BGR pixels OpenCV's imread
would return for them.  The expected pixels come from Pillow, which wraps the same libjpeg-turbo (islow IDCT,
fancy up-sampling) and zlib that OpenCV's imread uses; hand-built files (Adam7, 16-bit) carry hand-derived
expectations.  Run from the repo root:  python tests/golden/make_images.py
"""
import io
import os
import struct
import zlib

import numpy as np
from PIL import Image, ImageOps

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "images")


def smooth(h, w, seed, ch=3):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, ch))
    for c in range(ch):
        a, b, p, q = rng.uniform(0.02, 0.4, 4)
        img[..., c] = 127 + 70 * np.sin(a * xx + p * 9) * np.cos(b * yy + q * 9) + 40 * np.sin(0.9 * a * (xx + yy))
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def noise(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def expect_bgr(data, exif=False):
    im = Image.open(io.BytesIO(data))
    if exif:
        im = ImageOps.exif_transpose(im)
    return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


def png_chunk(t, body):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)


def write_png_raw(w, h, depth, ctype, interlace, passes):
    """passes: list of 2-D/3-D uint8 sample arrays already in scanline byte form per pass"""
    raw = b""
    for rows in passes:
        for r in rows:
            raw += b"\x00" + bytes(r)
    return (b"\x89PNG\r\n\x1a\n" + png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) +
            png_chunk(b"IDAT", zlib.compress(raw, 9)) + png_chunk(b"IEND", b""))


def main():
    os.makedirs(OUT, exist_ok=True)
    files, expected = {}, {}

    def add(name, data, exp):
        files[name] = data
        expected[name] = exp

    def jpeg(name, arr, **kw):
        buf = io.BytesIO()
        mode = "L" if arr.ndim == 2 else "RGB"
        Image.fromarray(arr, mode).save(buf, "JPEG", **kw)
        add(name, buf.getvalue(), expect_bgr(buf.getvalue(), exif="exif" in kw))

    jpeg("j444_q90.jpg", smooth(48, 64, 1), quality=90, subsampling=0)
    jpeg("j420_q75_odd.jpg", smooth(45, 67, 2), quality=75, subsampling=2)
    jpeg("j422_q85.jpg", smooth(33, 50, 3), quality=85, subsampling=1)
    jpeg("j420_noise_q30.jpg", noise(40, 56, 4), quality=30, subsampling=2)
    jpeg("j420_noise_q100.jpg", noise(24, 40, 5), quality=100, subsampling=2)
    jpeg("j420_w3.jpg", smooth(5, 3, 6), quality=90, subsampling=2)          # chroma width <= 2: no fancy up-sampling
    jpeg("j420_w5h2.jpg", smooth(2, 5, 7), quality=90, subsampling=2)
    jpeg("j422_w4.jpg", smooth(9, 4, 8), quality=90, subsampling=1)
    jpeg("j420_1x1.jpg", smooth(1, 1, 9), quality=90, subsampling=2)
    jpeg("jgray.jpg", smooth(31, 40, 10, ch=1)[..., 0], quality=80)
    jpeg("jprog420.jpg", smooth(47, 61, 11), quality=80, subsampling=2, progressive=True)
    jpeg("jprog444_noise.jpg", noise(33, 35, 12), quality=60, subsampling=0, progressive=True)
    jpeg("jprog_gray.jpg", smooth(20, 29, 13, ch=1)[..., 0], quality=70, progressive=True)
    jpeg("jopt420.jpg", smooth(40, 40, 14), quality=70, subsampling=2, optimize=True)
    jpeg("j420_160x120.jpg", smooth(120, 160, 15), quality=85, subsampling=2)
    for kw, nm in ((dict(restart_marker_blocks=3), "jrst_blocks.jpg"), (dict(restart_marker_rows=1), "jrst_rows.jpg")):
        try:
            jpeg(nm, smooth(50, 70, 16), quality=80, subsampling=2, **kw)
        except Exception as e:                                              # older Pillow
            print("skipped", nm, e)
    try:
        jpeg("jprog_rst.jpg", smooth(50, 70, 17), quality=80, subsampling=2, progressive=True, restart_marker_rows=1)
    except Exception as e:
        print("skipped jprog_rst", e)
    for sub in ("4:4:0", "4:1:1"):
        try:
            jpeg("j" + sub.replace(":", "") + ".jpg", smooth(37, 53, 18), quality=85, subsampling=sub)
        except Exception as e:
            print("skipped", sub, e)
    for orient in (2, 3, 4, 5, 6, 7, 8):
        ex = Image.Exif()
        ex[0x0112] = orient
        jpeg(f"jexif{orient}.jpg", smooth(21, 34, 20 + orient), quality=90, subsampling=2, exif=ex.tobytes())

    def png(name, im, **kw):
        buf = io.BytesIO()
        im.save(buf, "PNG", **kw)
        add(name, buf.getvalue(), expect_bgr(buf.getvalue()))

    rgb = smooth(29, 37, 30)
    png("p_rgb.png", Image.fromarray(rgb, "RGB"))
    rgba = np.dstack([rgb, noise(29, 37, 31)[..., 0]])
    png("p_rgba.png", Image.fromarray(rgba, "RGBA"))
    png("p_gray.png", Image.fromarray(rgb[..., 0], "L"))
    png("p_graya.png", Image.fromarray(rgba[..., 2:4].copy(), "LA"))
    png("p_pal.png", Image.fromarray(rgb, "RGB").quantize(37))
    png("p_pal4.png", Image.fromarray(rgb, "RGB").quantize(11), bits=4)
    png("p_pal2.png", Image.fromarray(rgb, "RGB").quantize(4), bits=2)
    png("p_bw.png", Image.fromarray(rgb[..., 1] > 127))
    # hand-built: Adam7 interlaced RGB8, 16-bit RGB, 4-bit grey
    h, w = 11, 13
    im = noise(h, w, 32)
    xs, ys, dx, dy = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
    passes = [[im[y, xs[p]::dx[p]].reshape(-1) for y in range(ys[p], h, dy[p])] for p in range(7) if xs[p] < w and ys[p] < h]
    add("p_adam7.png", write_png_raw(w, h, 8, 2, 1, passes), np.ascontiguousarray(im[:, :, ::-1]))
    hi, lo = noise(6, 9, 33), noise(6, 9, 34)
    rows16 = [np.stack([hi[y], lo[y]], -1).reshape(-1) for y in range(6)]
    add("p_rgb16.png", write_png_raw(9, 6, 16, 2, 0, [rows16]), np.ascontiguousarray(hi[:, :, ::-1]))
    g4 = np.random.default_rng(35).integers(0, 16, (5, 7), dtype=np.uint8)
    rows4 = [np.packbits(np.unpackbits(g4[y][:, None], axis=1)[:, 4:].reshape(-1)) for y in range(5)]
    add("p_gray4.png", write_png_raw(7, 5, 4, 0, 0, [rows4]), np.repeat((g4.astype(np.int32) * 255 // 15).astype(np.uint8)[..., None], 3, -1))

    buf = io.BytesIO(); Image.fromarray(rgb, "RGB").save(buf, "BMP"); add("b_rgb.bmp", buf.getvalue(), np.ascontiguousarray(rgb[:, :, ::-1]))
    buf = io.BytesIO(); Image.fromarray(rgb, "RGB").save(buf, "PPM"); add("n_rgb.ppm", buf.getvalue(), np.ascontiguousarray(rgb[:, :, ::-1]))
    buf = io.BytesIO(); Image.fromarray(rgb[..., 0], "L").save(buf, "PPM"); add("n_gray.pgm", buf.getvalue(), np.repeat(rgb[..., :1], 3, -1))

    # config C1 (BASELINE.json configs[0]): one 640x640 JPEG; its expected pixels are kept as a SHA-256 only
    import hashlib, json
    big = smooth(640, 640, 99)
    buf = io.BytesIO(); Image.fromarray(big, "RGB").save(buf, "JPEG", quality=80, subsampling=2)
    files["c1_640x640.jpg"] = buf.getvalue()
    with open(os.path.join(HERE, "images_sha256.json"), "w") as f:
        json.dump({"c1_640x640.jpg": hashlib.sha256(expect_bgr(buf.getvalue()).tobytes()).hexdigest()}, f, indent=1)

    for name, data in files.items():
        with open(os.path.join(OUT, name), "wb") as f:
            f.write(data)
    np.savez_compressed(os.path.join(HERE, "images_expected.npz"), **expected)
    print(len(files), "files,", sum(len(d) for d in files.values()), "bytes")


if __name__ == "__main__":
    main()
