"""Authors the small PNG / JPEG fixtures of tests/test_image_file.py (tests/golden/images/*) with encoders written here (zlib
for PNG; a baseline + progressive Huffman JPEG encoder, numpy DCT), decodes them with the REFERENCE's stb_image
(oracle/_ref/libref.so, ref_image_load = stbi_load(..., STBI_rgb_alpha)) and stores those outputs in ref_images.npz, plus
SHA-256 digests of stb's output for every texture the reference ships (tests compare when /root/reference is present).
Run in the build container: python tests/golden/make_images.py"""
import ctypes as C
import glob
import hashlib
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "images")
rng = np.random.default_rng(2024)


# ------------------------------------------------------------------------------------------------------------------ PNG
def png_chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)


def png_filter_rows(rows, bpp, filters):
    out = bytearray()
    prev = bytearray(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)]
        f = bytearray(len(row))
        for i in range(len(row)):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) >> 1
            else:
                pp = a + b - c
                pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            f[i] = (row[i] - p) & 255
        out.append(ft)
        out += f
        prev = bytearray(row)
    return bytes(out)


def pack_rows(samples, depth):
    """samples: (h, w, channels) integer array -> list of packed row byte strings"""
    h, w, ch = samples.shape
    rows = []
    for y in range(h):
        flat = samples[y].reshape(-1)
        if depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in flat))
        elif depth == 8:
            rows.append(bytes(int(v) for v in flat))
        else:
            bits = "".join(format(int(v), f"0{depth}b") for v in flat)
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


def write_png(name, samples, depth, ctype, interlace=0, palette=None, trns=None, filters=(0, 1, 2, 3, 4), level=9, strategy=0, split=1):
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    if interlace:
        xs, ys, dx, dy = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
        raw = b""
        for p in range(7):
            sub = samples[ys[p]::dy[p], xs[p]::dx[p]]
            if sub.shape[0] and sub.shape[1]:
                raw += png_filter_rows(pack_rows(sub, depth), bpp, filters)
    else:
        raw = png_filter_rows(pack_rows(samples, depth), bpp, filters)
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
    z = co.compress(raw) + co.flush()
    data = b"\x89PNG\r\n\x1a\n" + png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    if palette is not None:
        data += png_chunk(b"PLTE", bytes(int(v) for v in np.asarray(palette).reshape(-1)))
    if trns is not None:
        data += png_chunk(b"tRNS", trns)
    n = max(1, len(z) // split)
    for i in range(0, len(z), n):
        data += png_chunk(b"IDAT", z[i:i + n])
    data += png_chunk(b"tEXt", b"Comment\0authored by tests/golden/make_images.py") + png_chunk(b"IEND", b"")
    open(os.path.join(OUT, name), "wb").write(data)


def make_pngs():
    def img(h, w, ch, depth):
        x = rng.integers(0, 1 << depth, size=(h, w, ch))
        gy, gx = np.mgrid[0:h, 0:w]
        x[..., 0] = ((gx * 3 + gy * 5) * ((1 << depth) - 1) // (3 * w + 5 * h)) % (1 << depth)   # a ramp: exercises the predictors
        return x
    for d in (1, 2, 4, 8, 16):
        write_png(f"grey{d}.png", img(13, 19, 1, d), d, 0)
        write_png(f"grey{d}_adam7.png", img(11, 9, 1, d), d, 0, interlace=1)
    write_png("grey2_trns.png", img(9, 14, 1, 2), 2, 0, trns=struct.pack(">H", 2))
    write_png("grey16_trns.png", np.full((5, 6, 1), 0x1234) * (rng.random((5, 6, 1)) < 0.5) + 7, 16, 0, trns=struct.pack(">H", 0x1234 + 7))
    for d in (8, 16):
        write_png(f"rgb{d}.png", img(17, 12, 3, d), d, 2)
        write_png(f"rgb{d}_adam7.png", img(10, 15, 3, d), d, 2, interlace=1)
        write_png(f"greya{d}.png", img(8, 11, 2, d), d, 4)
        write_png(f"rgba{d}.png", img(12, 12, 4, d), d, 6, interlace=d == 16)
    t = img(6, 7, 3, 8); t[2, 3] = (10, 20, 30); t[4, 1] = (10, 20, 30)
    write_png("rgb8_trns.png", t, 8, 2, trns=struct.pack(">HHH", 10, 20, 30))
    for d in (1, 2, 4, 8):
        n = 1 << d
        pal = rng.integers(0, 256, size=(n, 3))
        write_png(f"pal{d}.png", rng.integers(0, n, size=(9, 13, 1)), d, 3, palette=pal, trns=bytes(rng.integers(0, 256, size=n // 2 + 1).tolist()))
    write_png("pal8_adam7_noalpha.png", rng.integers(0, 200, size=(14, 10, 1)), 8, 3, interlace=1, palette=rng.integers(0, 256, size=(200, 3)))
    big = img(64, 96, 3, 8)
    write_png("rgb8_stored.png", big, 8, 2, level=0)
    write_png("rgb8_fixed.png", big, 8, 2, strategy=zlib.Z_FIXED)
    write_png("rgb8_multi_idat.png", big, 8, 2, level=6, split=7, filters=(4,))
    write_png("rgb8_1x1.png", img(1, 1, 3, 8), 8, 2)


# ----------------------------------------------------------------------------------------------------------------- JPEG
ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57,
      50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
QL = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56,
      68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99]
QC = [17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32


class Bits:
    def __init__(self):
        self.out = bytearray(); self.acc = 0; self.n = 0

    def put(self, v, n):
        for i in range(n - 1, -1, -1):
            self.acc = (self.acc << 1) | ((v >> i) & 1); self.n += 1
            if self.n == 8:
                self.out.append(self.acc)
                if self.acc == 0xff: self.out.append(0)
                self.acc = 0; self.n = 0

    def flush(self):
        while self.n: self.put(1, 1)
        b = bytes(self.out); self.out = bytearray()
        return b


def mag(v):
    a = abs(v); s = a.bit_length()
    return s, (v if v >= 0 else v + (1 << s) - 1)


class Huff:
    """equal-length canonical code over the symbols in use (never the all-ones code)"""
    def __init__(self, symbols):
        self.sym = sorted(set(symbols)) or [0]
        self.len = max(1, len(self.sym).bit_length())
        self.code = {s: i for i, s in enumerate(self.sym)}

    def dht(self, tc, th):
        counts = [0] * 16; counts[self.len - 1] = len(self.sym)
        return bytes([tc << 4 | th] + counts + self.sym)

    def put(self, bits, s):
        bits.put(self.code[s], self.len)


def seg(m, d):
    return bytes([0xff, m]) + struct.pack(">H", len(d) + 2) + d


def jpeg_planes(rgb, sampling, grey):
    h, w, _ = rgb.shape
    r, g, b = (rgb[..., k].astype(np.float64) for k in range(3))
    y = 0.299 * r + 0.587 * g + 0.114 * b
    if grey: return [(y, 1, 1)]
    cb = -0.168736 * r - 0.331264 * g + 0.5 * b + 128
    cr = 0.5 * r - 0.418688 * g - 0.081312 * b + 128
    hs, vs = sampling
    def sub(p):
        ph, pw = -(-h // vs) * vs, -(-w // hs) * hs
        q = np.pad(p, ((0, ph - h), (0, pw - w)), mode="edge")
        return q.reshape(ph // vs, vs, pw // hs, hs).mean(axis=(1, 3))
    return [(y, hs, vs), (sub(cb), 1, 1), (sub(cr), 1, 1)]


def quantised_blocks(plane, bw, bh, q):
    ph, pw = bh * 8, bw * 8
    p = np.pad(plane, ((0, ph - plane.shape[0]), (0, pw - plane.shape[1])), mode="edge") - 128.0
    k = np.arange(8)
    Cm = np.sqrt(2 / 8) * np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16); Cm[0] /= np.sqrt(2)
    out = np.zeros((bh, bw, 64), np.int64)
    qn = np.array(q, np.float64).reshape(8, 8)
    for by in range(bh):
        for bx in range(bw):
            d = Cm @ p[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] @ Cm.T
            out[by, bx] = np.rint(d / qn).astype(np.int64).reshape(64)[ZZ]   # stored in zigzag order
    return out


def write_jpeg(name, rgb, sampling=(1, 1), grey=False, restart=0, progressive=False, q16=False, quality_scale=1.0, jfif=True):
    h, w, _ = rgb.shape
    planes = jpeg_planes(rgb, sampling, grey)
    hmax, vmax = max(p[1] for p in planes), max(p[2] for p in planes)
    mx, my = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    qt = [[max(1, min(65535 if q16 else 255, int(v * quality_scale * (300 if q16 else 1)))) for v in t] for t in (QL, QC)]
    comps = []
    for i, (pl, ch, cv) in enumerate(planes):
        tq = 0 if i == 0 else 1
        comps.append(dict(id=i + 1, h=ch, v=cv, tq=tq, blocks=quantised_blocks(pl, mx * ch, my * cv, qt[tq]),
                          cw=-(-w * ch // hmax), chh=-(-h * cv // vmax)))
    data = b"\xff\xd8"
    if jfif: data += seg(0xe0, b"JFIF\0\1\1\0\0\1\0\1\0\0")
    data += seg(0xfe, b"authored by tests/golden/make_images.py")
    for t in range(1 if grey else 2):
        data += seg(0xdb, bytes([(1 if q16 else 0) << 4 | t]) + (b"".join(struct.pack(">H", qt[t][z]) for z in ZZ) if q16 else bytes(qt[t][z] for z in ZZ)))
    data += seg(0xc2 if progressive else 0xc0, struct.pack(">BHHB", 8, h, w, len(comps)) + b"".join(bytes([c["id"], c["h"] << 4 | c["v"], c["tq"]]) for c in comps))
    if restart: data += seg(0xdd, struct.pack(">H", restart))

    def mcu_blocks(cs):
        """(component, bx, by) in scan order for an interleaved scan over `cs`, grouped per MCU"""
        if len(cs) == 1:
            c = cs[0]
            return [[(c, bx, by)] for by in range(-(-c["chh"] // 8)) for bx in range(-(-c["cw"] // 8))]
        return [[(c, x * c["h"] + i, y * c["v"] + j) for c in cs for j in range(c["v"]) for i in range(c["h"])] for y in range(my) for x in range(mx)]

    def run_scan(cs, ss, se, ah, al, coder):
        """two passes: collect the symbols, then emit with a table built from them"""
        nonlocal data
        results = []
        for emit in (False, True):
            syms_dc, syms_ac = [], []
            if emit:
                hd, ha = Huff(results[0]), Huff(results[1])
                tables = b""
                if ss == 0 and ah == 0: tables += hd.dht(0, 0)
                if se > 0: tables += ha.dht(1, 0)
                if tables: data += seg(0xc4, tables)
                data += seg(0xda, bytes([len(cs)]) + b"".join(bytes([c["id"], 0]) for c in cs) + bytes([ss, se, ah << 4 | al]))
            bits = Bits()
            pred = {c["id"]: 0 for c in cs}
            for k, mcu in enumerate(mcu_blocks(cs)):
                if restart and k and k % restart == 0:
                    if emit: data += bits.flush() + bytes([0xff, 0xd0 + (k // restart - 1) % 8])
                    pred = {c["id"]: 0 for c in cs}
                for c, bx, by in mcu:
                    coder(c["blocks"][by, bx], pred, c["id"], bits if emit else None, syms_dc, syms_ac, hd if emit else None, ha if emit else None)
            if emit: data += bits.flush()
            results = [syms_dc, syms_ac]

    def code_sequential(blk, pred, cid, bits, sd, sa, hd, ha, ss=0, se=63, al=0, dc=True):
        if dc:
            v = int(blk[0]) >> al if al else int(blk[0])
            s, m = mag(v - pred[cid]); pred[cid] = v
            sd.append(s)
            if bits: hd.put(bits, s); bits.put(m, s)
        if se == 0: return
        run = 0
        last = max([k for k in range(max(ss, 1), se + 1) if (abs(int(blk[k])) >> al)] or [0])
        for k in range(max(ss, 1), se + 1):
            v = int(blk[k]); v = (abs(v) >> al) * (1 if v >= 0 else -1)
            if k > last: break
            if v == 0: run += 1; continue
            while run > 15:
                sa.append(0xf0)
                if bits: ha.put(bits, 0xf0)
                run -= 16
            s, m = mag(v)
            sa.append(run << 4 | s)
            if bits: ha.put(bits, run << 4 | s); bits.put(m, s)
            run = 0
        if last < se:
            sa.append(0)
            if bits: ha.put(bits, 0)

    if not progressive:
        run_scan(comps, 0, 63, 0, 0, code_sequential)
    else:
        def dc_first(blk, pred, cid, bits, sd, sa, hd, ha): code_sequential(blk, pred, cid, bits, sd, sa, hd, ha, 0, 0, 1, True)
        def dc_refine(blk, pred, cid, bits, sd, sa, hd, ha):
            if bits: bits.put(int(blk[0]) & 1, 1)
        def ac_first(ss, se, al):
            return lambda blk, pred, cid, bits, sd, sa, hd, ha: code_sequential(blk, pred, cid, bits, sd, sa, hd, ha, ss, se, al, False)
        def ac_refine(ss, se, al):
            def f(blk, pred, cid, bits, sd, sa, hd, ha):
                absv = [abs(int(blk[k])) >> al for k in range(64)]
                eob = max([k for k in range(ss, se + 1) if absv[k] == 1] or [-1])
                run, pending = 0, []
                for k in range(ss, se + 1):
                    t = absv[k]
                    if t == 0: run += 1; continue
                    while run > 15 and k <= eob:
                        sa.append(0xf0)
                        if bits:
                            ha.put(bits, 0xf0)
                            for b in pending: bits.put(b, 1)
                        pending = []; run -= 16
                    if t > 1: pending.append(t & 1); continue
                    sa.append(run << 4 | 1)
                    if bits:
                        ha.put(bits, run << 4 | 1); bits.put(0 if int(blk[k]) < 0 else 1, 1)
                        for b in pending: bits.put(b, 1)
                    pending = []; run = 0
                if run > 0 or pending:
                    sa.append(0)          # EOB0: the band ends in this block; its correction bits follow
                    if bits:
                        ha.put(bits, 0)
                        for b in pending: bits.put(b, 1)
            return f
        run_scan(comps, 0, 0, 0, 1, dc_first)
        for c in comps: run_scan([c], 1, 5, 0, 2, ac_first(1, 5, 2))
        for c in comps: run_scan([c], 6, 63, 0, 2, ac_first(6, 63, 2))
        for c in comps: run_scan([c], 1, 63, 2, 1, ac_refine(1, 63, 1))
        run_scan(comps, 0, 0, 1, 0, dc_refine)
        for c in comps: run_scan([c], 1, 63, 1, 0, ac_refine(1, 63, 0))
    data += b"\xff\xd9"
    open(os.path.join(OUT, name), "wb").write(data)


def make_jpegs():
    def picture(h, w):
        gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
        r = 128 + 100 * np.sin(gx / 5.0) * np.cos(gy / 7.0)
        g = 255 * gx / max(1, w - 1)
        b = 255 * ((gx // 6 + gy // 4) % 2)
        img = np.stack([r, g, b], -1) + rng.normal(0, 12, size=(h, w, 3))
        return np.clip(img, 0, 255)
    write_jpeg("j444.jpg", picture(37, 29))
    write_jpeg("j420.jpg", picture(43, 51), sampling=(2, 2))
    write_jpeg("j420_tiny.jpg", picture(3, 2), sampling=(2, 2))
    write_jpeg("j420_w1.jpg", picture(17, 1), sampling=(2, 2))
    write_jpeg("j422.jpg", picture(24, 33), sampling=(2, 1))
    write_jpeg("j440.jpg", picture(33, 24), sampling=(1, 2))
    write_jpeg("j411.jpg", picture(20, 40), sampling=(4, 1))
    write_jpeg("jgrey.jpg", picture(30, 30), grey=True)
    write_jpeg("j420_rst.jpg", picture(40, 56), sampling=(2, 2), restart=3)
    write_jpeg("j444_q16.jpg", picture(16, 24), q16=True)
    write_jpeg("j444_sharp.jpg", picture(32, 32), quality_scale=0.08)     # fine quantisation: large coefficients, clamping
    write_jpeg("j444_nojfif.jpg", picture(16, 16), jfif=False)
    write_jpeg("jp444.jpg", picture(35, 41), progressive=True)
    write_jpeg("jp420.jpg", picture(48, 40), sampling=(2, 2), progressive=True, quality_scale=0.3)
    write_jpeg("jpgrey_rst.jpg", picture(26, 31), grey=True, progressive=True, restart=5)


def main():
    os.makedirs(OUT, exist_ok=True)
    for f in glob.glob(os.path.join(OUT, "*")): os.remove(f)
    make_pngs()
    make_jpegs()
    from oracle import binding as ob
    r = ob.ref_lib()
    assert r is not None, "build oracle/_ref first: make -C oracle ref"

    def stb(path):
        w, h = C.c_int(), C.c_int()
        if r.ref_image_load(path.encode(), C.byref(w), C.byref(h), None, 0) != 0: return None
        px = np.zeros((h.value, w.value, 4), np.uint8)
        r.ref_image_load(path.encode(), C.byref(w), C.byref(h), C.c_void_p(px.ctypes.data), px.nbytes)
        return px
    data = {}
    for f in sorted(os.listdir(OUT)):
        px = stb(os.path.join(OUT, f))
        assert px is not None, f"stb_image rejects the authored fixture {f}"
        data[f] = px
    shipped = {}
    for f in sorted(glob.glob("/root/reference/src/data/house/textures/*")):
        if f.lower().endswith((".jpg", ".jpeg", ".png")):
            px = stb(f)
            shipped[os.path.basename(f)] = np.frombuffer(hashlib.sha256(np.array(px.shape, np.int64).tobytes() + px.tobytes()).digest(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_images.npz"), **{"fixture/" + k: v for k, v in data.items()},
                        **{"shipped/" + k: v for k, v in shipped.items()})
    print(len(data), "fixtures,", len(shipped), "digests of shipped textures;", sum(os.path.getsize(os.path.join(OUT, f)) for f in data), "bytes of fixtures")


if __name__ == "__main__":
    main()
