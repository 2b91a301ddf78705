// Texture files of the reference's scenes -> RGBA8 (row f1).  The reference reads textures with
// stbi_load(path, &w, &h, &c, STBI_rgb_alpha) (OptiXPathTracer/scene_shift.cpp:35-40, vendored stb_image v2.27); its shipped
// `house` scene uses baseline and progressive JPEG (4:4:4, 4:2:0, greyscale) and 8-bit RGB PNG.  This is a decoder of those
// formats written from the format specifications (ITU T.81, RFC 1950/1951, the PNG specification) whose OUTPUT is held
// bit-exactly to stb_image's by tests/test_image_file.py (oracle/_ref links the reference's own stb_image.cpp):
//   PNG   every colour type and bit depth, tRNS, Adam7; 16-bit samples keep their high byte; lossless, so any correct
//         decoder agrees -- the conventions that matter are the expansion of sub-byte grey (x 255 / (2^d - 1)) and alpha
//   JPEG  Huffman baseline / extended / progressive, 1 or 3 components, restart intervals.  Lossy: the arithmetic after
//         entropy decoding decides the bytes, so it follows what stb_image computes: the "islow" 13-bit fixed-point inverse
//         DCT (2 extra bits kept after the column pass, +128 level shift folded into the row pass), triangle-filter chroma
//         upsampling centred as JFIF prescribes, and the 12.8 fixed-point YCbCr conversion
// Binary PPM (P6) stays in scene_file.cpp.  Host code, no GPU.
#include <cstdint>
#include <new>
#include <stdexcept>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spcbpt.h"
#include "scene_file.h"

namespace {

typedef std::vector<uint8_t> Bytes;

bool read_file(const std::string& path, Bytes& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0 || n > (1l << 30)) { fclose(f); return false; }
    out.resize((size_t)n);
    const bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

// ============================================================================================================ inflate
// RFC 1951 DEFLATE inside an RFC 1950 zlib stream.  Canonical Huffman decoding by code length (count / first-code walk).
struct Inflater {
    const uint8_t* p; const uint8_t* end;
    uint32_t bits = 0; int nbits = 0;
    bool bad = false;
    size_t max_out = (size_t)1 << 31;   // the caller knows how many bytes the image needs: a stream that inflates to more is refused
    int bit() {
        if (nbits == 0) { if (p >= end) { bad = true; return 0; } bits = *p++; nbits = 8; }
        const int b = bits & 1; bits >>= 1; nbits--; return b;
    }
    uint32_t take(int n) { uint32_t v = 0; for (int i = 0; i < n; i++) v |= (uint32_t)bit() << i; return v; }
    struct Code { uint16_t count[16]; uint16_t symbol[320]; };
    static bool build(Code& c, const uint8_t* len, int n) {
        memset(c.count, 0, sizeof(c.count));
        for (int i = 0; i < n; i++) c.count[len[i]]++;
        c.count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; l++) { left = (left << 1) - c.count[l]; if (left < 0) return false; }
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + c.count[l];
        for (int i = 0; i < n; i++) if (len[i]) c.symbol[offs[len[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(const Code& c) {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= bit();
            const int cnt = c.count[l];
            if (code - cnt < first) return c.symbol[index + (code - first)];
            index += cnt; first += cnt; first <<= 1; code <<= 1;
        }
        bad = true;
        return 0;
    }
    bool run(Bytes& out) {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        if (end - p < 2 || (p[0] & 15) != 8 || ((p[0] << 8) | p[1]) % 31 != 0 || (p[1] & 32)) return false;  // zlib header, no preset dictionary
        p += 2;
        int final_block;
        do {
            final_block = bit();
            const int type = (int)take(2);
            if (type == 0) {
                nbits = 0;
                if (end - p < 4) return false;
                const uint32_t len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
                p += 4;
                if ((len ^ 0xffffu) != nlen || (size_t)(end - p) < len) return false;
                out.insert(out.end(), p, p + len);
                p += len;
            } else if (type == 1 || type == 2) {
                Code lc, dc;
                uint8_t lens[320];
                if (type == 1) {
                    for (int i = 0; i < 144; i++) lens[i] = 8;
                    for (int i = 144; i < 256; i++) lens[i] = 9;
                    for (int i = 256; i < 280; i++) lens[i] = 7;
                    for (int i = 280; i < 288; i++) lens[i] = 8;
                    build(lc, lens, 288);
                    for (int i = 0; i < 30; i++) lens[i] = 5;
                    build(dc, lens, 30);
                } else {
                    const int nl = (int)take(5) + 257, nd = (int)take(5) + 1, nc = (int)take(4) + 4;
                    uint8_t cl[19] = {0};
                    for (int i = 0; i < nc; i++) cl[order[i]] = (uint8_t)take(3);
                    Code cc;
                    if (nl > 286 || nd > 30 || !build(cc, cl, 19)) return false;
                    int i = 0;
                    while (i < nl + nd && !bad) {
                        const int sym = decode(cc);
                        if (sym < 16) lens[i++] = (uint8_t)sym;
                        else {
                            int rep, val = 0;
                            if (sym == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int)take(2); }
                            else if (sym == 17) rep = 3 + (int)take(3);
                            else rep = 11 + (int)take(7);
                            if (i + rep > nl + nd) return false;
                            while (rep--) lens[i++] = (uint8_t)val;
                        }
                    }
                    if (bad || !build(lc, lens, nl) || !build(dc, lens + nl, nd)) return false;
                }
                while (!bad) {
                    const int sym = decode(lc);
                    if (out.size() > max_out) return false;
                    if (sym < 256) out.push_back((uint8_t)sym);
                    else if (sym == 256) break;
                    else {
                        if (sym > 285) return false;
                        const int len = lbase[sym - 257] + (int)take(lext[sym - 257]);
                        const int ds = decode(dc);
                        if (ds > 29) return false;
                        const size_t dist = dbase[ds] + take(dext[ds]);
                        if (dist > out.size()) return false;
                        const size_t from = out.size() - dist;
                        for (int k = 0; k < len; k++) out.push_back(out[from + k]);
                    }
                }
            } else return false;
        } while (!final_block && !bad);
        return !bad;
    }
};

// ================================================================================================================ PNG
bool decode_png(const Bytes& file, Bytes& rgba, int& W, int& H) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    Bytes idat, plte;
    uint8_t pal_alpha[256];
    memset(pal_alpha, 255, sizeof(pal_alpha));
    bool have_trns = false, have_hdr = false;
    uint16_t trns[3] = {0, 0, 0};
    while (pos + 8 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        const uint8_t* data = &file[pos + 8];
        if (len > file.size() || pos + 12 + len > file.size()) return false;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) return false;
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28) || data[10] != 0 || data[11] != 0 || interlace > 1) return false;
            have_hdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!memcmp(type, "tRNS", 4)) {
            have_trns = true;
            if (ctype == 3) { for (uint32_t i = 0; i < len && i < 256; i++) pal_alpha[i] = data[i]; }
            else if (ctype == 0 && len >= 2) trns[0] = (uint16_t)be16(data);
            else if (ctype == 2 && len >= 6) { for (int k = 0; k < 3; k++) trns[k] = (uint16_t)be16(data + 2 * k); }
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_hdr) return false;
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || !(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) return false;
    if ((ctype == 3 && depth == 16) || ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8)) return false;
    Bytes raw;
    Inflater inf;
    inf.p = idat.data(); inf.end = idat.data() + idat.size();
    inf.max_out = ((size_t)w * channels * depth / 8 + 9) * ((size_t)h + 7 * (interlace ? 1 : 0)) + 1024;   // every pass row: filter byte + samples
    if (!inf.run(raw)) return false;
    // samples of the whole image as 16-bit values (channel-interleaved), filled pass by pass
    std::vector<uint16_t> samp((size_t)w * h * channels);
    static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
    const int passes = interlace ? 7 : 1;
    size_t off = 0;
    const int bpp_bits = channels * depth, bpp = (bpp_bits + 7) / 8;
    for (int ps = 0; ps < passes; ps++) {
        const uint32_t pw = interlace ? (w - xs[ps] + dx[ps] - 1) / dx[ps] : w, ph = interlace ? (h - ys[ps] + dy[ps] - 1) / dy[ps] : h;
        if (pw == 0 || ph == 0) continue;
        const size_t stride = ((size_t)pw * bpp_bits + 7) / 8;
        if (off + (stride + 1) * ph > raw.size()) return false;
        Bytes prev(stride, 0), cur(stride);
        for (uint32_t y = 0; y < ph; y++) {
            const uint8_t ft = raw[off];
            const uint8_t* src = &raw[off + 1];
            off += stride + 1;
            for (size_t i = 0; i < stride; i++) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
                int pred = 0;
                switch (ft) {
                    case 0: pred = 0; break;
                    case 1: pred = a; break;
                    case 2: pred = b; break;
                    case 3: pred = (a + b) >> 1; break;
                    case 4: { const int pp = a + b - c, pa = pp > a ? pp - a : a - pp, pb = pp > b ? pp - b : b - pp, pc = pp > c ? pp - c : c - pp;
                              pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                    default: return false;
                }
                cur[i] = (uint8_t)(src[i] + pred);
            }
            const uint32_t oy = interlace ? ys[ps] + y * dy[ps] : y;
            for (uint32_t x = 0; x < pw; x++) {
                const uint32_t ox = interlace ? xs[ps] + x * dx[ps] : x;
                uint16_t* dst = &samp[((size_t)oy * w + ox) * channels];
                for (int ch = 0; ch < channels; ch++) {
                    if (depth == 16) dst[ch] = (uint16_t)be16(&cur[((size_t)x * channels + ch) * 2]);
                    else if (depth == 8) dst[ch] = cur[(size_t)x * channels + ch];
                    else { const size_t bitpos = (size_t)x * depth; dst[ch] = (cur[bitpos >> 3] >> (8 - depth - (bitpos & 7))) & ((1 << depth) - 1); }
                }
            }
            prev.swap(cur);
        }
    }
    W = (int)w; H = (int)h;
    rgba.resize((size_t)w * h * 4);
    const int maxv = (1 << depth) - 1;
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint16_t* s = &samp[i * channels];
        uint8_t* o = &rgba[i * 4];
        auto to8 = [&](uint16_t v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : depth == 8 ? (uint8_t)v : (uint8_t)(v * (255 / maxv)); };
        if (ctype == 3) {
            const size_t k = s[0];
            if (k * 3 + 2 < plte.size()) { o[0] = plte[k * 3]; o[1] = plte[k * 3 + 1]; o[2] = plte[k * 3 + 2]; o[3] = pal_alpha[k & 255]; }
            else { o[0] = o[1] = o[2] = 0; o[3] = 255; }
        } else if (ctype == 0) {
            o[0] = o[1] = o[2] = to8(s[0]);
            o[3] = (have_trns && s[0] == trns[0]) ? 0 : 255;
        } else if (ctype == 4) {
            o[0] = o[1] = o[2] = to8(s[0]); o[3] = to8(s[1]);
        } else if (ctype == 2) {
            o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]);
            o[3] = (have_trns && s[0] == trns[0] && s[1] == trns[1] && s[2] == trns[2]) ? 0 : 255;
        } else {
            o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]); o[3] = to8(s[3]);
        }
    }
    return true;
}

// =============================================================================================================== JPEG
const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                             28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                             47, 55, 62, 63};

struct JHuff {  // canonical code: per length the first code, the last + 1 and the index of its first symbol
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
};
struct JComp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;      // padded to whole MCUs
    int w = 0, h_px = 0;                 // sample extent of the component inside the image (ceil)
    std::vector<int16_t> coef;           // blocks_w * blocks_h * 64, natural order
    Bytes plane;                         // blocks_w*8 x blocks_h*8 samples
    int dc_pred = 0;
};

struct JpegDecoder {
    const uint8_t* p; const uint8_t* end;
    uint16_t quant[4][64];
    JHuff hdc[4], hac[4];
    JComp comp[3];
    int ncomp = 0, width = 0, height = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    bool progressive = false, jfif = false;
    int adobe_transform = -1;
    int restart_interval = 0;
    // entropy-coded segment reader
    uint32_t acc = 0; int nacc = 0;
    bool hit_marker = false;
    int eobrun = 0;
    bool bad = false;

    int next_bit() {
        if (nacc == 0) {
            uint8_t b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xff) {
                    uint8_t b2 = p < end ? *p : 0;
                    while (b2 == 0xff && p + 1 < end) { p++; b2 = *p; }   // fill bytes
                    if (b2 == 0) p++;                                    // stuffed zero: a data byte 0xff
                    else { hit_marker = true; p--; b = 0; }              // a marker: feed zeros from here on
                }
            }
            acc = b; nacc = 8;
        }
        nacc--;
        return (acc >> nacc) & 1;
    }
    int receive(int n) { int v = 0; while (n--) v = (v << 1) | next_bit(); return v; }
    static int extend(int v, int n) { return n == 0 ? 0 : (v < (1 << (n - 1)) ? v - (1 << n) + 1 : v); }
    int decode(const JHuff& t) {
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | next_bit();
            if (t.maxcode[l] >= 0 && code < t.maxcode[l] && code >= t.mincode[l]) return t.vals[t.valptr[l] + code - t.mincode[l]];
        }
        bad = true;
        return 0;
    }
    void reset_entropy() { acc = 0; nacc = 0; hit_marker = false; eobrun = 0; for (int i = 0; i < ncomp; i++) comp[i].dc_pred = 0; }

    bool parse_dht(const uint8_t* d, int len) {
        while (len >= 17) {
            const int tc = d[0] >> 4, th = d[0] & 15;
            if (tc > 1 || th > 3) return false;
            JHuff& t = tc ? hac[th] : hdc[th];
            int total = 0, code = 0, k = 0;
            for (int l = 1; l <= 16; l++) {
                const int n = d[l];
                t.valptr[l] = k; t.mincode[l] = code;
                code += n; k += n; total += n;
                t.maxcode[l] = n ? code : -1;
                code <<= 1;
            }
            if (total > 256 || len < 17 + total) return false;
            memcpy(t.vals, d + 17, (size_t)total);
            t.present = true;
            d += 17 + total; len -= 17 + total;
        }
        return len == 0;
    }
    bool parse_dqt(const uint8_t* d, int len) {
        while (len > 0) {
            const int pq = d[0] >> 4, tq = d[0] & 15;
            if (tq > 3 || pq > 1 || len < 1 + 64 * (pq + 1)) return false;
            for (int i = 0; i < 64; i++) quant[tq][kZigzag[i]] = pq ? (uint16_t)be16(d + 1 + 2 * i) : d[1 + i];
            d += 1 + 64 * (pq + 1); len -= 1 + 64 * (pq + 1);
        }
        return true;
    }
    bool parse_sof(const uint8_t* d, int len) {
        if (len < 6 || d[0] != 8) return false;
        height = (int)be16(d + 1); width = (int)be16(d + 3); ncomp = d[5];
        if ((ncomp != 1 && ncomp != 3) || width < 1 || height < 1 || len < 6 + 3 * ncomp || (long long)width * height > (1ll << 28)) return false;
        hmax = vmax = 1;
        for (int i = 0; i < ncomp; i++) {
            comp[i].id = d[6 + 3 * i]; comp[i].h = d[7 + 3 * i] >> 4; comp[i].v = d[7 + 3 * i] & 15; comp[i].tq = d[8 + 3 * i];
            if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) return false;
            if (comp[i].h > hmax) hmax = comp[i].h;
            if (comp[i].v > vmax) vmax = comp[i].v;
        }
        for (int i = 0; i < ncomp; i++) if (hmax % comp[i].h || vmax % comp[i].v) return false;
        mcus_x = (width + 8 * hmax - 1) / (8 * hmax); mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; i++) {
            JComp& c = comp[i];
            c.blocks_w = mcus_x * c.h; c.blocks_h = mcus_y * c.v;
            c.w = (width * c.h + hmax - 1) / hmax; c.h_px = (height * c.v + vmax - 1) / vmax;
            c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
        }
        return true;
    }

    // one 8x8 block of a sequential scan: coefficients dequantised on the fly, stored as 16-bit values (wrapping like a short)
    void block_sequential(JComp& c, int16_t* b) {
        const int t = decode(hdc[c.td]);
        const int diff = t ? extend(receive(t), t) : 0;
        c.dc_pred += diff;
        b[0] = (int16_t)(c.dc_pred * quant[c.tq][0]);
        for (int k = 1; k < 64;) {
            const int rs = decode(hac[c.ta]), r = rs >> 4, s = rs & 15;
            if (s == 0) { if (r != 15) break; k += 16; continue; }
            k += r;
            if (k > 63) { bad = true; break; }
            const int z = kZigzag[k++];
            b[z] = (int16_t)(extend(receive(s), s) * quant[c.tq][z]);
        }
    }
    void block_dc_progressive(JComp& c, int16_t* b, int ah, int al) {
        if (ah == 0) {
            const int t = decode(hdc[c.td]);
            const int diff = t ? extend(receive(t), t) : 0;
            c.dc_pred += diff;
            b[0] = (int16_t)(c.dc_pred * (1 << al));
        } else if (next_bit()) b[0] = (int16_t)(b[0] + (1 << al));
    }
    void block_ac_progressive(JComp& c, int16_t* b, int ss, int se, int ah, int al) {
        if (ah == 0) {
            if (eobrun) { eobrun--; return; }
            for (int k = ss; k <= se;) {
                const int rs = decode(hac[c.ta]), r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += receive(r); break; }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) { bad = true; break; }
                    b[kZigzag[k++]] = (int16_t)(extend(receive(s), s) * (1 << al));
                }
            }
            return;
        }
        const int bit = 1 << al;
        auto refine = [&](int16_t& v) { if (next_bit() && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit); };
        int k = ss;
        if (eobrun == 0) {
            while (k <= se) {
                const int rs = decode(hac[c.ta]), s = rs & 15;
                int r = rs >> 4, value = 0;
                if (s == 0) {
                    if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += receive(r); eobrun++; break; }   // this band included
                } else {
                    if (s != 1) { bad = true; return; }
                    value = next_bit() ? bit : -bit;
                }
                while (k <= se) {   // skip r zero-history coefficients, refining the non-zero ones on the way
                    int16_t& v = b[kZigzag[k++]];
                    if (v != 0) refine(v);
                    else { if (r == 0) { if (value) v = (int16_t)value; break; } r--; }
                }
                if (bad) return;
            }
        }
        if (eobrun) {
            for (; k <= se; k++) { int16_t& v = b[kZigzag[k]]; if (v != 0) refine(v); }
            eobrun--;
        }
    }

    bool scan(const uint8_t* hdr, int len) {
        if (len < 1) return false;
        const int ns = hdr[0];
        if (ns < 1 || ns > ncomp || len < 4 + 2 * ns) return false;
        int idx[3];
        for (int i = 0; i < ns; i++) {
            int k = -1;
            for (int j = 0; j < ncomp; j++) if (comp[j].id == hdr[1 + 2 * i]) k = j;
            if (k < 0) return false;
            idx[i] = k;
            comp[k].td = hdr[2 + 2 * i] >> 4; comp[k].ta = hdr[2 + 2 * i] & 15;
            if (comp[k].td > 3 || comp[k].ta > 3) return false;
        }
        const int ss = hdr[1 + 2 * ns], se = hdr[2 + 2 * ns], ah = hdr[3 + 2 * ns] >> 4, al = hdr[3 + 2 * ns] & 15;
        if (progressive) { if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13 || (ss == 0 && se != 0) || (ss != 0 && ns != 1)) return false; }
        else if (ss != 0 || se != 63 || ah != 0 || al != 0) return false;
        reset_entropy();
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        auto restart_check = [&]() -> bool {   // after every MCU: an RSTn marker is due when the interval is used up
            if (--todo > 0) return true;
            if (!hit_marker) {   // the marker has not been reached by the bit reader yet: it is the next thing in the stream
                while (p < end && *p != 0xff) p++;
            }
            while (p + 1 < end && p[0] == 0xff && p[1] == 0xff) p++;
            if (p + 1 < end && p[0] == 0xff && p[1] >= 0xd0 && p[1] <= 0xd7) {
                p += 2;
                reset_entropy();
                todo = restart_interval;
                return true;
            }
            return false;   // no restart marker: the scan ends here (EOI or the next segment)
        };
        auto one_block = [&](JComp& c, int bx, int by) {
            int16_t* b = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
            if (!progressive) block_sequential(c, b);
            else if (ss == 0) block_dc_progressive(c, b, ah, al);
            else block_ac_progressive(c, b, ss, se, ah, al);
        };
        bool more = true;
        if (ns == 1) {   // non-interleaved: the component's own blocks, only those that cover the image
            JComp& c = comp[idx[0]];
            const int bw = (c.w + 7) / 8, bh = (c.h_px + 7) / 8;
            for (int by = 0; by < bh && more && !bad; by++)
                for (int bx = 0; bx < bw && more && !bad; bx++) { one_block(c, bx, by); more = restart_check(); }
        } else {
            for (int my = 0; my < mcus_y && more && !bad; my++)
                for (int mx = 0; mx < mcus_x && more && !bad; mx++) {
                    for (int i = 0; i < ns; i++) {
                        JComp& c = comp[idx[i]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) one_block(c, mx * c.h + x, my * c.v + y);
                    }
                    more = restart_check();
                }
        }
        // position the reader on the marker that ended the entropy-coded data
        if (!hit_marker) { while (p + 1 < end && !(p[0] == 0xff && p[1] != 0 && p[1] != 0xff && !(p[1] >= 0xd0 && p[1] <= 0xd7))) p++; }
        return !bad;
    }

    // 8x8 inverse DCT on dequantised coefficients.  The decoded bytes must be the reference's (its vendored stb_image, itself the
    // IJG "islow" integer transform), so the INTEGERS are fixed: the Loeffler-Ligtenberg-Moschytz factorisation with 12-bit
    // constants, a column pass that keeps two extra bits (rounding 512, shift 10) and a row pass that folds the +128 level shift
    // and the rounding into one bias (shift 17).  Integer sums are exact, so the formulation is free; this one writes the 1-D
    // transform as an even half E[0..3] (inputs 0, 2, 4, 6) and an odd half O[0..3] (inputs 1, 3, 5, 7) with the outputs
    // y[k] = E[k] + O[k], y[7 - k] = E[k] - O[k], each odd output as its own dot product over shared pair sums.
    struct Lane8 { int even[4], odd[4]; };
    static int fix12(double x) { return (int)(x * 4096 + 0.5); }
    static Lane8 idct_1d(int x0, int x1, int x2, int x3, int x4, int x5, int x6, int x7) {
        // rotation constants: sqrt(2) cos(k pi / 16) combinations of the LLM flow graph
        static const int r6 = fix12(0.5411961f), r2m6 = fix12(0.765366865f), r2p6 = fix12(-1.847759065f), r3 = fix12(1.175875602f);
        static const int k7 = fix12(0.298631336f), k5 = fix12(2.053119869f), k3 = fix12(3.072711026f), k1 = fix12(1.501321110f);
        static const int m17 = fix12(-0.899976223f), m35 = fix12(-2.562915447f), m37 = fix12(-1.961570560f), m15 = fix12(-0.390180644f);
        Lane8 y;
        // even half: a rotation of (x2, x6) around the butterfly of (x0, x4)
        const int rot = (x2 + x6) * r6;
        const int lo = rot + x2 * r2m6, hi = rot + x6 * r2p6;
        const int sum = (x0 + x4) * 4096, dif = (x0 - x4) * 4096;
        y.even[0] = sum + lo; y.even[3] = sum - lo;
        y.even[1] = dif + hi; y.even[2] = dif - hi;
        // odd half: pair sums shared between the four outputs
        const int s17 = x1 + x7, s35 = x3 + x5, s37 = x3 + x7, s15 = x1 + x5;
        const int all = (s37 + s15) * r3;
        const int a17 = all + s17 * m17, a35 = all + s35 * m35, b37 = s37 * m37, b15 = s15 * m15;
        y.odd[0] = x1 * k1 + a17 + b15;
        y.odd[1] = x3 * k3 + a35 + b37;
        y.odd[2] = x5 * k5 + a35 + b15;
        y.odd[3] = x7 * k7 + a17 + b37;
        return y;
    }
    static void idct(const int16_t* d, uint8_t* out, int stride) {
        int mid[64];   // after the column pass, scaled by 4
        for (int col = 0; col < 8; col++) {
            const int16_t* s = d + col;
            if (!(s[8] | s[16] | s[24] | s[32] | s[40] | s[48] | s[56])) {   // a DC-only column is flat: (4096 dc + 512) >> 10 = 4 dc
                for (int k = 0; k < 8; k++) mid[col + 8 * k] = s[0] * 4;
                continue;
            }
            const Lane8 y = idct_1d(s[0], s[8], s[16], s[24], s[32], s[40], s[48], s[56]);
            for (int k = 0; k < 4; k++) {
                mid[col + 8 * k] = (y.even[k] + y.odd[k] + 512) >> 10;
                mid[col + 8 * (7 - k)] = (y.even[k] - y.odd[k] + 512) >> 10;
            }
        }
        const int bias = 65536 + (128 << 17);   // rounding of the final shift by 17 + the level shift, in one constant
        for (int row = 0; row < 8; row++) {
            const int* s = mid + 8 * row;
            uint8_t* o = out + (size_t)row * stride;
            const Lane8 y = idct_1d(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
            for (int k = 0; k < 4; k++) {
                const int a = (y.even[k] + y.odd[k] + bias) >> 17, b = (y.even[k] - y.odd[k] + bias) >> 17;
                o[k] = (uint8_t)(a < 0 ? 0 : a > 255 ? 255 : a);
                o[7 - k] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
            }
        }
    }

    bool run(const Bytes& file) {
        p = file.data(); end = p + file.size();
        if (file.size() < 4 || p[0] != 0xff || p[1] != 0xd8) return false;
        p += 2;
        memset(quant, 0, sizeof(quant));
        bool have_sof = false, done = false;
        while (!done && p + 4 <= end) {
            if (*p != 0xff) { p++; continue; }
            while (p < end && *p == 0xff) p++;
            if (p >= end) break;
            const int m = *p++;
            if (m == 0xd9) { done = true; break; }
            if (m == 0x01 || (m >= 0xd0 && m <= 0xd7) || m == 0) continue;
            if (p + 2 > end) return false;
            const int len = (int)be16(p) - 2;
            const uint8_t* d = p + 2;
            if (len < 0 || d + len > end) return false;
            p = d + len;
            if (m == 0xc0 || m == 0xc1 || m == 0xc2) { progressive = m == 0xc2; if (have_sof || !parse_sof(d, len)) return false; have_sof = true; }
            else if (m == 0xc4) { if (!parse_dht(d, len)) return false; }
            else if (m == 0xdb) { if (!parse_dqt(d, len)) return false; }
            else if (m == 0xdd) { if (len < 2) return false; restart_interval = (int)be16(d); }
            else if (m == 0xe0) { if (len >= 5 && !memcmp(d, "JFIF\0", 5)) jfif = true; }
            else if (m == 0xee) { if (len >= 12 && !memcmp(d, "Adobe\0", 6)) adobe_transform = d[11]; }
            else if (m == 0xda) { if (!have_sof || !scan(d, len)) return false; }
            else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) return false;  // lossless / hierarchical / arithmetic
        }
        if (!have_sof) return false;
        for (int i = 0; i < ncomp; i++) {
            JComp& c = comp[i];
            const int stride = c.blocks_w * 8;
            c.plane.assign((size_t)stride * c.blocks_h * 8, 0);
            for (int by = 0; by < c.blocks_h; by++)
                for (int bx = 0; bx < c.blocks_w; bx++) {
                    int16_t* b = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
                    if (progressive) for (int k = 0; k < 64; k++) b[k] = (int16_t)(b[k] * quant[c.tq][k]);
                    idct(b, &c.plane[(size_t)by * 8 * stride + (size_t)bx * 8], stride);
                }
        }
        return true;
    }

    // One output row of a component at full resolution.  2x factors use the triangle filter (3/4 near + 1/4 far, in each
    // direction), co-sited as JFIF centres chroma; other factors replicate.
    void upsampled_row(const JComp& c, int y, uint8_t* out, std::vector<int>& tmp) const {
        const int hs = hmax / c.h, vs = vmax / c.v, stride = c.blocks_w * 8;
        const int wl = (width + hs - 1) / hs;
        const uint8_t *near_row, *far_row;
        if (vs == 2) {
            const int cy = y >> 1, last = c.h_px - 1;
            auto row = [&](int r) { return &c.plane[(size_t)(r < 0 ? 0 : r > last ? last : r) * stride]; };
            if (y == 0) { near_row = row(0); far_row = row(0); }
            else if (y & 1) { near_row = row(cy); far_row = row(cy + 1); }
            else { near_row = row(cy); far_row = row(cy - 1); }
        } else {
            int r = vs == 1 ? y : y / vs;
            if (r > c.h_px - 1) r = c.h_px - 1;
            near_row = far_row = &c.plane[(size_t)r * stride];
        }
        if (hs == 1 && vs == 1) { memcpy(out, near_row, (size_t)width); return; }
        if (hs == 1 && vs == 2) { for (int i = 0; i < width; i++) out[i] = (uint8_t)((3 * near_row[i] + far_row[i] + 2) >> 2); return; }
        std::vector<uint8_t> wide((size_t)2 * wl + 4);
        if (hs == 2 && vs == 1) {
            const uint8_t* in = near_row;
            if (wl == 1) wide[0] = wide[1] = in[0];
            else {
                wide[0] = in[0];
                wide[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                for (int i = 1; i < wl - 1; i++) {
                    const int n = 3 * in[i] + 2;
                    wide[2 * i] = (uint8_t)((n + in[i - 1]) >> 2);
                    wide[2 * i + 1] = (uint8_t)((n + in[i + 1]) >> 2);
                }
                wide[2 * (wl - 1)] = (uint8_t)((in[wl - 2] * 3 + in[wl - 1] + 2) >> 2);
                wide[2 * (wl - 1) + 1] = in[wl - 1];
            }
            memcpy(out, wide.data(), (size_t)width);
            return;
        }
        if (hs == 2 && vs == 2) {
            tmp.resize((size_t)wl);
            for (int i = 0; i < wl; i++) tmp[i] = 3 * near_row[i] + far_row[i];
            if (wl == 1) wide[0] = wide[1] = (uint8_t)((tmp[0] + 2) >> 2);
            else {
                wide[0] = (uint8_t)((tmp[0] + 2) >> 2);
                for (int i = 1; i < wl; i++) {
                    wide[2 * i - 1] = (uint8_t)((3 * tmp[i - 1] + tmp[i] + 8) >> 4);
                    wide[2 * i] = (uint8_t)((3 * tmp[i] + tmp[i - 1] + 8) >> 4);
                }
                wide[2 * wl - 1] = (uint8_t)((tmp[wl - 1] + 2) >> 2);
            }
            memcpy(out, wide.data(), (size_t)width);
            return;
        }
        for (int i = 0; i < width; i++) out[i] = near_row[i / hs];   // other factors: replication
    }

    void to_rgba(Bytes& rgba) const {
        rgba.resize((size_t)width * height * 4);
        std::vector<uint8_t> r0((size_t)width + 8), r1((size_t)width + 8), r2((size_t)width + 8);
        std::vector<int> tmp;
        // stored as RGB when the component ids spell "RGB", or an Adobe marker says "no transform" and there is no JFIF header
        const bool is_rgb = ncomp == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') || (adobe_transform == 0 && !jfif));
        auto fixed = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
        const int k_rcr = fixed(1.40200f), k_gcr = fixed(0.71414f), k_gcb = fixed(0.34414f), k_bcb = fixed(1.77200f);
        auto clamp = [](int x) -> uint8_t { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); };
        for (int y = 0; y < height; y++) {
            uint8_t* o = &rgba[(size_t)y * width * 4];
            upsampled_row(comp[0], y, r0.data(), tmp);
            if (ncomp == 1) {
                for (int x = 0; x < width; x++) { o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = r0[x]; o[4 * x + 3] = 255; }
                continue;
            }
            upsampled_row(comp[1], y, r1.data(), tmp);
            upsampled_row(comp[2], y, r2.data(), tmp);
            for (int x = 0; x < width; x++) {
                if (is_rgb) { o[4 * x] = r0[x]; o[4 * x + 1] = r1[x]; o[4 * x + 2] = r2[x]; o[4 * x + 3] = 255; continue; }
                const int yf = (r0[x] << 20) + (1 << 19), cr = r2[x] - 128, cb = r1[x] - 128;
                const int r = yf + cr * k_rcr;
                const int g = yf + (cr * -k_gcr) + (int)((uint32_t)(cb * -k_gcb) & 0xffff0000u);
                const int b = yf + cb * k_bcb;
                o[4 * x] = clamp(r >> 20); o[4 * x + 1] = clamp(g >> 20); o[4 * x + 2] = clamp(b >> 20); o[4 * x + 3] = 255;
            }
        }
    }
};

bool decode_jpeg(const Bytes& file, Bytes& rgba, int& w, int& h) {
    std::vector<JpegDecoder> d(1);   // on the heap: the tables are a few KB
    if (!d[0].run(file)) return false;
    w = d[0].width; h = d[0].height;
    d[0].to_rgba(rgba);
    return true;
}

}  // namespace

namespace spc_loader {
// Any texture file the reference's scenes may name: JPEG, PNG or binary PPM, chosen by the file's first bytes.
bool load_image(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h) {
    Bytes file;
    if (!read_file(path, file) || file.size() < 4) return false;
    if (file[0] == 0xff && file[1] == 0xd8) return decode_jpeg(file, rgba, w, h);
    if (file[0] == 0x89 && file[1] == 'P') return decode_png(file, rgba, w, h);
    if (file[0] == 'P' && file[1] == '6') return load_ppm(path, rgba, w, h);
    return false;
}
}  // namespace spc_loader

static int image_load_impl(const char* path, int* width, int* height, uint8_t* rgba, size_t capacity_bytes) {
    if (!path || !width || !height) return SPCBPT_ERR_INVALID_ARG;
    std::vector<uint8_t> px;
    int w = 0, h = 0;
    if (!spc_loader::load_image(path, px, w, h)) return SPCBPT_ERR_IO;
    *width = w; *height = h;
    if (rgba) {
        if (capacity_bytes < px.size()) return SPCBPT_ERR_CAPACITY;
        memcpy(rgba, px.data(), px.size());
    }
    return SPCBPT_OK;
}

extern "C" int spcbpt_image_load(const char* path, int* width, int* height, uint8_t* rgba, size_t capacity_bytes) {
    try { return image_load_impl(path, width, height, rgba, capacity_bytes); }   // never throw across the C ABI
    catch (const std::bad_alloc&) { return SPCBPT_ERR_CAPACITY; }
    catch (const std::exception&) { return SPCBPT_ERR_IO; }
}
