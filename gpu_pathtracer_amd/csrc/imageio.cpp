// imageio.cpp — image files on the host side of the path.
//
// What the reference does with stb_image / stb_image_write / tinyexr (all
// third-party, not copied):
//   ImageIO::SavePng     src/imageio.cpp:61-78   flip Y, clamp to [0,1], truncate to 8 bit
//   ImageIO::LoadTexture src/imageio.cpp:11-59   flip Y, sRGB -> linear with powf(x, 2.2f)
//   Texture::Texture     src/texture.h:15-27     quantise the linear value back to uchar4
//   ImageIO::LoadExr     src/imageio.cpp:80-102  lat-long environment map
//
// Here: a PNG writer (stored-deflate, no compression needed for a checker
// output), a PNG reader (every colour type and bit depth, Adam7 interlacing,
// colour keys) on a small inflate, and PFM (little- or big-endian float32)
// for linear radiance and environment maps, an OpenEXR scanline reader (NONE / RLE / ZIPS / ZIP / PIZ,
// half / float) for the reference's environment maps, and a baseline + progressive JPEG reader for textures.
#include "imageio.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <initializer_list>
#include <string>

#include "../../include/gpt.h"
#include "host_util.h"

namespace {

// ---- checksums -------------------------------------------------------------------
uint32_t crc32_update(uint32_t crc, const unsigned char *p, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}
uint32_t adler32(const unsigned char *p, size_t n)
{
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; ++i) {
        a = (a + p[i]) % 65521u;
        b = (b + a) % 65521u;
    }
    return (b << 16) | a;
}
void put_be32(std::vector<unsigned char> &v, uint32_t x)
{
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
void put_chunk(std::vector<unsigned char> &out, const char type[4], const std::vector<unsigned char> &data)
{
    put_be32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32_update(0, out.data() + start, out.size() - start));
}

// ---- inflate (RFC 1951) ------------------------------------------------------------
struct BitReader {
    const unsigned char *p;
    size_t n, pos = 0;
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool fail = false;
    int bits(int need)
    {
        while (bitcnt < need) {
            if (pos >= n) { fail = true; return 0; }
            bitbuf |= (uint32_t)p[pos++] << bitcnt;
            bitcnt += 8;
        }
        int v = (int)(bitbuf & ((1u << need) - 1));
        bitbuf >>= need;
        bitcnt -= need;
        return v;
    }
};
struct Huffman {
    short count[16];
    short symbol[320];
    void build(const unsigned char *lengths, int n)
    {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        short offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (short)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (short)i;
    }
    int decode(BitReader &br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= br.bits(1);
            if (br.fail) return -1;
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};
// `limit`: the most bytes the caller can use; a stream that inflates beyond it is refused (crafted files must not
// exhaust memory)
bool inflate_raw(const unsigned char *src, size_t n, std::vector<unsigned char> &out, size_t limit)
{
    static const short lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const short lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const short dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const short dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    BitReader br{src, n};
    int last;
    do {
        last = br.bits(1);
        int type = br.bits(2);
        if (br.fail) return false;
        if (type == 0) {
            br.bitbuf = 0; br.bitcnt = 0;
            if (br.pos + 4 > n) return false;
            unsigned len = src[br.pos] | (src[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > n) return false;
            if (out.size() + len > limit) return false;
            out.insert(out.end(), src + br.pos, src + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            unsigned char lengths[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lengths[i] = 8;
                for (; i < 256; ++i) lengths[i] = 9;
                for (; i < 280; ++i) lengths[i] = 7;
                for (; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for (i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                static const unsigned char order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                int nlen = br.bits(5) + 257, ndist = br.bits(5) + 1, ncode = br.bits(4) + 4;
                if (br.fail || nlen > 286 || ndist > 30) return false;
                unsigned char cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (unsigned char)br.bits(3);
                Huffman clh;
                clh.build(cl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = clh.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lengths[idx++] = (unsigned char)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (idx == 0) return false; val = lengths[idx - 1]; rep = 3 + br.bits(2); }
                        else if (sym == 17) rep = 3 + br.bits(3);
                        else rep = 11 + br.bits(7);
                        if (idx + rep > nlen + ndist) return false;
                        while (rep--) lengths[idx++] = (unsigned char)val;
                    }
                }
                lit.build(lengths, nlen);
                dist.build(lengths + nlen, ndist);
            }
            for (;;) {
                int sym = lit.decode(br);
                if (sym < 0 || br.fail) return false;
                if (sym < 256) { if (out.size() >= limit) return false; out.push_back((unsigned char)sym); }
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return false;
                    int len = lbase[sym] + br.bits(lext[sym]);
                    int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    size_t d = (size_t)dbase[ds] + (size_t)br.bits(dext[ds]);
                    if (d > out.size() || out.size() + (size_t)len > limit) return false;
                    size_t from = out.size() - d;
                    for (int k = 0; k < len; ++k) out.push_back(out[from + (size_t)k]);
                }
            }
        } else {
            return false;
        }
    } while (!last);
    return !br.fail;
}

bool read_file(const char *path, std::vector<unsigned char> &data)
{
    FILE *f = gpt_fopen_read(path);
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    data.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? std::fread(data.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    return got == data.size();
}
uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

namespace imageio {

bool write_png_rgb8(const char *path, int width, int height, const unsigned char *rgb_top_down)
{
    std::vector<unsigned char> raw;
    raw.reserve((size_t)height * ((size_t)width * 3 + 1));
    for (int y = 0; y < height; ++y) {
        raw.push_back(0);   // filter: none
        raw.insert(raw.end(), rgb_top_down + (size_t)y * width * 3, rgb_top_down + (size_t)(y + 1) * width * 3);
    }
    std::vector<unsigned char> z;
    z.push_back(0x78); z.push_back(0x01);
    size_t pos = 0;
    do {
        size_t n = raw.size() - pos;
        if (n > 65535) n = 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xff)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xff)); z.push_back((unsigned char)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        pos += n;
    } while (pos < raw.size());
    put_be32(z, adler32(raw.data(), raw.size()));

    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, (uint32_t)width);
    put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", {});
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    size_t w = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return w == out.size();
}

// PNG -> RGBA8, rows top-down: every colour type and bit depth of the format (1 / 2 / 4 / 8 / 16 bits; grey, RGB, palette, with
// alpha), Adam7 interlacing, tRNS colour keys.  The choices the format leaves to a reader follow stb_image, the reference's
// (tests/test_imageio_reference.py compares with stb_image itself): 16-bit samples keep their HIGH byte, 1 / 2 / 4-bit grey is
// scaled by 255 / 85 / 17, a colour key is compared at full sample precision and adds an alpha channel.
// components = what stb_image reports for the file: 1, 2, 3 or 4 (palette -> 3, with tRNS 4; grey / RGB with a key -> 2 / 4)
bool read_png(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba)
{
    std::vector<unsigned char> d;
    if (!read_file(path, d) || d.size() < 33 || std::memcmp(d.data(), "\x89PNG\r\n\x1a\n", 8) != 0) return false;
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte, trns;
    bool have_trns = false;
    width = height = 0;
    while (pos + 12 <= d.size()) {
        uint32_t len = be32(&d[pos]);
        const unsigned char *type = &d[pos + 4];
        if (len > d.size() || pos + 12 + len > d.size()) return false;
        const unsigned char *body = &d[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13 || width) return false;
            width = (int)be32(body); height = (int)be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0) return false;                  // compression / filter method
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) { trns.assign(body, body + len); have_trns = true; }
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    if (width <= 0 || height <= 0 || width > (1 << 24) || height > (1 << 24) || interlace > 1 || idat.size() < 6) return false;
    if ((uint64_t)width * (uint64_t)height > ((uint64_t)1 << 28)) return false;
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!depth_ok) return false;
    if (ctype == 3 && (plte.empty() || plte.size() > 768)) return false;
    // the sub-images the data holds: the whole picture, or the seven Adam7 passes
    struct Pass { int x0, y0, dx, dy, w, h; };
    std::vector<Pass> passes;
    if (!interlace) passes.push_back(Pass{0, 0, 1, 1, width, height});
    else {
        static const int x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int k = 0; k < 7; ++k) {
            const int w = (width - x0[k] + dx[k] - 1) / dx[k], h = (height - y0[k] + dy[k] - 1) / dy[k];
            if (w > 0 && h > 0) passes.push_back(Pass{x0[k], y0[k], dx[k], dy[k], w, h});
        }
    }
    auto row_bytes = [&](int w) { return ((size_t)w * (size_t)ch * (size_t)depth + 7) >> 3; };
    size_t expect = 0;
    for (const Pass &ps : passes) expect += (row_bytes(ps.w) + 1) * (size_t)ps.h;
    std::vector<unsigned char> raw;
    if (!inflate_raw(idat.data() + 2, idat.size() - 2, raw, expect) || raw.size() < expect) return false;
    const size_t bpp = (size_t)(ch * depth >= 8 ? ch * depth / 8 : 1);         // the filters' "corresponding byte of the pixel to the left"
    std::vector<uint16_t> samp((size_t)width * height * ch);                   // every sample at its own precision
    std::vector<unsigned char> cur_row, up_row;
    size_t at = 0;
    for (const Pass &ps : passes) {
        const size_t rb = row_bytes(ps.w);
        up_row.assign(rb, 0);
        cur_row.assign(rb, 0);
        for (int j = 0; j < ps.h; ++j) {
            const unsigned char *in = &raw[at];
            at += rb + 1;
            const int ft = in[0];
            if (ft > 4) return false;
            for (size_t i = 0; i < rb; ++i) {
                const int a = i >= bpp ? cur_row[i - bpp] : 0, b = up_row[i], c = i >= bpp ? up_row[i - bpp] : 0, x = in[i + 1];
                cur_row[i] = (unsigned char)(ft == 0 ? x : ft == 1 ? x + a : ft == 2 ? x + b : ft == 3 ? x + ((a + b) >> 1) : x + paeth(a, b, c));
            }
            const size_t y = (size_t)ps.y0 + (size_t)j * ps.dy;
            for (int i = 0; i < ps.w; ++i) {
                uint16_t *o = &samp[(y * width + (size_t)ps.x0 + (size_t)i * ps.dx) * ch];
                for (int c = 0; c < ch; ++c) {
                    const size_t k = (size_t)i * ch + c;                       // sample number in the row
                    if (depth == 16) o[c] = (uint16_t)(cur_row[2 * k] << 8 | cur_row[2 * k + 1]);
                    else if (depth == 8) o[c] = cur_row[k];
                    else {
                        const size_t bit = k * (size_t)depth;
                        o[c] = (uint16_t)((cur_row[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1 << depth) - 1));
                    }
                }
            }
            cur_row.swap(up_row);
        }
    }
    const int scale = depth == 1 ? 0xff : depth == 2 ? 0x55 : depth == 4 ? 0x11 : 1;
    auto to8 = [&](uint16_t v) { return (unsigned char)(depth == 16 ? v >> 8 : v * scale); };
    // a colour key (tRNS of a grey or RGB file): 16-bit files compare the whole sample, the others the 8-bit value
    uint16_t key[3] = {0, 0, 0};
    const bool keyed = have_trns && (ctype == 0 || ctype == 2);
    if (keyed) {
        if (trns.size() < (size_t)(2 * ch)) return false;
        for (int c = 0; c < ch; ++c) {
            const uint16_t v = (uint16_t)(trns[2 * c] << 8 | trns[2 * c + 1]);
            key[c] = depth == 16 ? v : (uint16_t)((v & 255) * scale);
        }
    }
    components = ctype == 3 ? (have_trns ? 4 : 3) : ch + (keyed ? 1 : 0);
    rgba.resize((size_t)width * height * 4);
    for (size_t i = 0; i < (size_t)width * height; ++i) {
        const uint16_t *v = &samp[i * ch];
        unsigned char r, g, b, a = 255;
        if (ctype == 3) {
            const size_t k = v[0];
            if (3 * k + 2 >= plte.size()) return false;
            r = plte[3 * k]; g = plte[3 * k + 1]; b = plte[3 * k + 2];
            if (k < trns.size()) a = trns[k];
        } else if (ctype == 0 || ctype == 4) {
            r = g = b = to8(v[0]);
            if (ctype == 4) a = to8(v[1]);
            else if (keyed && (depth == 16 ? v[0] : (uint16_t)r) == key[0]) a = 0;
        } else {
            r = to8(v[0]); g = to8(v[1]); b = to8(v[2]);
            if (ctype == 6) a = to8(v[3]);
            else if (keyed && (depth == 16 ? (v[0] == key[0] && v[1] == key[1] && v[2] == key[2]) : (r == key[0] && g == key[1] && b == key[2]))) a = 0;
        }
        rgba[4 * i] = r; rgba[4 * i + 1] = g; rgba[4 * i + 2] = b; rgba[4 * i + 3] = a;
    }
    return true;
}

// ---- JPEG: baseline and progressive DCT (Huffman, 8-bit, 1 or 3 components, any h/v sampling, restart
// intervals; spectral selection and successive approximation for SOF2 files).  The inverse DCT, the chroma up-sampling and the
// YCbCr -> RGB conversion restate stb_image's integer arithmetic (the reference's reader), so the bytes are the reference's
// bytes (tests/test_imageio_reference.py compares them with stb_image itself).  Four-component files (Adobe CMYK / YCCK) come out as RGB the way stb_image converts them.
// Arithmetic-coded, lossless and hierarchical files are refused.
namespace {
struct JpegHuff {
    unsigned char bits[17] = {0};
    unsigned char vals[256] = {0};
    int mincode[17] = {0}, maxcode[18], valptr[17] = {0};
    JpegHuff() { for (int &m : maxcode) m = -1; }          // a table the file never defines decodes nothing
    void build()
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};
struct JpegBits {
    const unsigned char *p;
    size_t n, pos;
    uint32_t buf = 0;
    int cnt = 0;
    bool eof = false;
    int bit()
    {
        if (cnt == 0) {
            unsigned char c = 0;
            if (pos < n) {
                c = p[pos++];
                if (c == 0xff) {
                    if (pos < n && p[pos] == 0) ++pos;
                    else { --pos; c = 0; eof = true; }     // a marker: feed zeros
                }
            } else eof = true;
            buf = c;
            cnt = 8;
        }
        --cnt;
        return (buf >> cnt) & 1;
    }
    int receive(int s)
    {
        int v = 0;
        for (int i = 0; i < s; ++i) v = (v << 1) | bit();
        return v;
    }
    int decode(const JpegHuff &h)
    {
        int code = 0;
        for (int l = 1; l <= 16; ++l) {
            code = (code << 1) | bit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        return -1;
    }
    void reset() { cnt = 0; eof = false; }
};
int jpeg_extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
// stb_image's inverse DCT (the reference's JPEG reader, include/stb/stb_image.h: a fixed-point factorisation with 12-bit
// constants; columns keep 2 extra bits, rows round once at the end).  A texel of a JPEG texture is DEFINED by this
// arithmetic for the reference, so it is restated here operation for operation, not approximated.
inline int jfix(double x) { return (int)(x * 4096.0 + 0.5); }
// (32-bit two's-complement arithmetic that wraps, written with unsigned operands: coefficients of a damaged file can overflow it)
typedef uint32_t jw;
struct JpegIdct1D { jw x0, x1, x2, x3, t0, t1, t2, t3; };
inline JpegIdct1D jpeg_idct_1d(jw s0, jw s1, jw s2, jw s3, jw s4, jw s5, jw s6, jw s7)
{
    static const jw c0541 = (jw)jfix(0.5411961f), cm1847 = (jw)jfix(-1.847759065f), c0765 = (jw)jfix(0.765366865f), c1175 = (jw)jfix(1.175875602f),
                    c0298 = (jw)jfix(0.298631336f), c2053 = (jw)jfix(2.053119869f), c3072 = (jw)jfix(3.072711026f), c1501 = (jw)jfix(1.501321110f),
                    cm0899 = (jw)jfix(-0.899976223f), cm2562 = (jw)jfix(-2.562915447f), cm1961 = (jw)jfix(-1.961570560f), cm0390 = (jw)jfix(-0.390180644f);
    JpegIdct1D r;
    jw p1 = (s2 + s6) * c0541;
    const jw e2 = p1 + s6 * cm1847, e3 = p1 + s2 * c0765;
    const jw e0 = (s0 + s4) * 4096u, e1 = (s0 - s4) * 4096u;
    r.x0 = e0 + e3; r.x3 = e0 - e3; r.x1 = e1 + e2; r.x2 = e1 - e2;
    jw t0 = s7, t1 = s5, t2 = s3, t3 = s1;
    jw p3 = t0 + t2, p4 = t1 + t3, p2 = t1 + t2;
    p1 = t0 + t3;
    const jw p5 = (p3 + p4) * c1175;
    t0 *= c0298; t1 *= c2053; t2 *= c3072; t3 *= c1501;
    p1 = p5 + p1 * cm0899;
    p2 = p5 + p2 * cm2562;
    p3 *= cm1961;
    p4 *= cm0390;
    r.t3 = t3 + p1 + p4; r.t2 = t2 + p2 + p3; r.t1 = t1 + p2 + p4; r.t0 = t0 + p1 + p3;
    return r;
}
inline unsigned char jclamp(int x) { return (unsigned char)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
inline int jsar(jw x, int n) { return (int32_t)x >> n; }          // arithmetic shift of the signed value
void jpeg_idct(const short in[64], unsigned char *out, int stride)
{
    jw v[64];
    for (int i = 0; i < 8; ++i) {
        const short *d = in + i;
        if (!d[8] && !d[16] && !d[24] && !d[32] && !d[40] && !d[48] && !d[56]) {       // a column with only its DC term
            const jw dc = (jw)(d[0] * 4);
            for (int k = 0; k < 8; ++k) v[k * 8 + i] = dc;
            continue;
        }
        JpegIdct1D r = jpeg_idct_1d((jw)d[0], (jw)d[8], (jw)d[16], (jw)d[24], (jw)d[32], (jw)d[40], (jw)d[48], (jw)d[56]);
        r.x0 += 512; r.x1 += 512; r.x2 += 512; r.x3 += 512;
        v[i] = (jw)jsar(r.x0 + r.t3, 10);       v[56 + i] = (jw)jsar(r.x0 - r.t3, 10);
        v[8 + i] = (jw)jsar(r.x1 + r.t2, 10);   v[48 + i] = (jw)jsar(r.x1 - r.t2, 10);
        v[16 + i] = (jw)jsar(r.x2 + r.t1, 10);  v[40 + i] = (jw)jsar(r.x2 - r.t1, 10);
        v[24 + i] = (jw)jsar(r.x3 + r.t0, 10);  v[32 + i] = (jw)jsar(r.x3 - r.t0, 10);
    }
    for (int i = 0; i < 8; ++i) {
        const jw *w = v + i * 8;
        unsigned char *o = out + (size_t)i * stride;
        JpegIdct1D r = jpeg_idct_1d(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
        const jw bias = 65536u + (128u << 17);      // rounding, and the level shift back to 0..255
        r.x0 += bias; r.x1 += bias; r.x2 += bias; r.x3 += bias;
        o[0] = jclamp(jsar(r.x0 + r.t3, 17)); o[7] = jclamp(jsar(r.x0 - r.t3, 17));
        o[1] = jclamp(jsar(r.x1 + r.t2, 17)); o[6] = jclamp(jsar(r.x1 - r.t2, 17));
        o[2] = jclamp(jsar(r.x2 + r.t1, 17)); o[5] = jclamp(jsar(r.x2 - r.t1, 17));
        o[3] = jclamp(jsar(r.x3 + r.t0, 17)); o[4] = jclamp(jsar(r.x3 - r.t0, 17));
    }
}
// stb_image's chroma up-sampling: one output row of `w` low-resolution samples widened by `hs`, from the nearer and the
// farther of the two source rows around it (triangle filters for the factors 2; replication for anything else)
void jpeg_upsample_row(unsigned char *out, const unsigned char *near_row, const unsigned char *far_row, int w, int hs, int vs)
{
    if (hs == 1 && vs == 1) { std::memcpy(out, near_row, (size_t)w); return; }
    if (hs == 1 && vs == 2) { for (int i = 0; i < w; ++i) out[i] = (unsigned char)((3 * near_row[i] + far_row[i] + 2) >> 2); return; }
    if (hs == 2 && vs == 1) {
        const unsigned char *in = near_row;
        if (w == 1) { out[0] = out[1] = in[0]; return; }
        out[0] = in[0];
        out[1] = (unsigned char)((in[0] * 3 + in[1] + 2) >> 2);
        for (int i = 1; i < w - 1; ++i) {
            const int n = 3 * in[i] + 2;
            out[2 * i] = (unsigned char)((n + in[i - 1]) >> 2);
            out[2 * i + 1] = (unsigned char)((n + in[i + 1]) >> 2);
        }
        out[2 * (w - 1)] = (unsigned char)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[2 * (w - 1) + 1] = in[w - 1];
        return;
    }
    if (hs == 2 && vs == 2) {
        int t1 = 3 * near_row[0] + far_row[0];
        if (w == 1) { out[0] = out[1] = (unsigned char)((t1 + 2) >> 2); return; }
        out[0] = (unsigned char)((t1 + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            const int t0 = t1;
            t1 = 3 * near_row[i] + far_row[i];
            out[2 * i - 1] = (unsigned char)((3 * t0 + t1 + 8) >> 4);
            out[2 * i] = (unsigned char)((3 * t1 + t0 + 8) >> 4);
        }
        out[2 * w - 1] = (unsigned char)((t1 + 2) >> 2);
        return;
    }
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < hs; ++j) out[i * hs + j] = near_row[i];
}
}  // namespace

bool read_jpeg(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba)
{
    static const unsigned char zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    std::vector<unsigned char> d;
    if (!read_file(path, d) || d.size() < 4 || d[0] != 0xff || d[1] != 0xd8) return false;
    uint16_t qt[4][64] = {{0}};
    bool jfif = false;
    int adobe_transform = -1;
    JpegHuff hdc[4], hac[4];
    // coefficients of every 8x8 block (natural order, not yet dequantised): a progressive file fills them in over
    // several scans (spectral selection Ss..Se, successive approximation Ah/Al), a baseline file in one
    struct Comp { int id, h, v, tq, td, ta, pred; std::vector<short> coef; int bw, bh, pw, ph; } comp[4];
    int ncomp = 0, hmax = 1, vmax = 1, restart = 0;
    bool progressive = false, have_scan = false;
    int mcux = 0, mcuy = 0;
    width = height = 0;
    size_t pos = 2;
    while (pos + 4 <= d.size()) {
        if (d[pos] != 0xff) { ++pos; continue; }
        const int marker = d[pos + 1];
        pos += 2;
        if (marker == 0xd8 || (marker >= 0xd0 && marker <= 0xd7) || marker == 0x01 || marker == 0xff) continue;
        if (marker == 0xd9) break;
        if (pos + 2 > d.size()) return false;
        const size_t len = (size_t)d[pos] << 8 | d[pos + 1];
        if (len < 2 || pos + len > d.size()) return false;
        const unsigned char *seg = &d[pos + 2];
        const size_t seglen = len - 2;
        if (marker == 0xe0 && seglen >= 5 && !std::memcmp(seg, "JFIF\0", 5)) jfif = true;
        else if (marker == 0xee && seglen >= 12 && !std::memcmp(seg, "Adobe\0", 6)) adobe_transform = seg[11];
        if (marker == 0xdb) {
            size_t q = 0;
            while (q < seglen) {
                const int pq = seg[q] >> 4, tq = seg[q] & 15;
                ++q;
                if (tq > 3 || q + (pq ? 128u : 64u) > seglen) return false;
                for (int i = 0; i < 64; ++i) { qt[tq][zigzag[i]] = (uint16_t)(pq ? seg[q] << 8 | seg[q + 1] : seg[q]); q += pq ? 2 : 1; }
            }
        } else if (marker == 0xc4) {
            size_t q = 0;
            while (q + 17 <= seglen) {
                const int tc = seg[q] >> 4, th = seg[q] & 15;
                if (th > 3) return false;
                JpegHuff &h = tc ? hac[th] : hdc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = seg[q + (size_t)l]; total += h.bits[l]; }
                q += 17;
                if (total > 256 || q + (size_t)total > seglen) return false;
                std::memcpy(h.vals, seg + q, (size_t)total);
                q += (size_t)total;
                h.build();
            }
        } else if (marker == 0xc0 || marker == 0xc1 || marker == 0xc2) {
            if (width || seglen < 6 || seg[0] != 8) return false;
            progressive = marker == 0xc2;
            height = seg[1] << 8 | seg[2];
            width = seg[3] << 8 | seg[4];
            ncomp = seg[5];
            if ((ncomp != 1 && ncomp != 3 && ncomp != 4) || seglen < 6 + 3 * (size_t)ncomp || width <= 0 || height <= 0) return false;
            if ((uint64_t)width * (uint64_t)height > ((uint64_t)1 << 27)) return false;      // (the coefficient store is 2 bytes per sample)
            for (int i = 0; i < ncomp; ++i) {
                comp[i].id = seg[6 + 3 * i];
                comp[i].h = seg[7 + 3 * i] >> 4;
                comp[i].v = seg[7 + 3 * i] & 15;
                comp[i].tq = seg[8 + 3 * i];
                if (comp[i].h < 1 || comp[i].v < 1 || comp[i].h > 4 || comp[i].v > 4 || comp[i].tq > 3) return false;
                if (comp[i].h > hmax) hmax = comp[i].h;
                if (comp[i].v > vmax) vmax = comp[i].v;
            }
            mcux = (width + 8 * hmax - 1) / (8 * hmax);
            mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            for (int i = 0; i < ncomp; ++i) {
                comp[i].bw = mcux * comp[i].h;            // blocks per row / column, padded to whole MCUs
                comp[i].bh = mcuy * comp[i].v;
                comp[i].pw = comp[i].bw * 8;
                comp[i].ph = comp[i].bh * 8;
                comp[i].coef.assign((size_t)comp[i].bw * comp[i].bh * 64, 0);
                comp[i].pred = 0;
                comp[i].td = comp[i].ta = 0;
            }
        } else if (marker >= 0xc5 && marker <= 0xcf && marker != 0xc8 && marker != 0xcc) {
            return false;    // lossless / hierarchical / arithmetic coding
        } else if (marker == 0xdd) {
            if (seglen < 2) return false;
            restart = seg[0] << 8 | seg[1];
        } else if (marker == 0xda) {
            if (!width || seglen < 1) return false;
            const int ns = seg[0];
            if (ns < 1 || ns > ncomp || seglen < 1 + 2 * (size_t)ns + 3) return false;
            int order[4];
            for (int i = 0; i < ns; ++i) {
                const int cid = seg[1 + 2 * i];
                int k = 0;
                while (k < ncomp && comp[k].id != cid) ++k;
                if (k == ncomp) return false;
                comp[k].td = seg[2 + 2 * i] >> 4;
                comp[k].ta = seg[2 + 2 * i] & 15;
                if (comp[k].td > 3 || comp[k].ta > 3) return false;
                order[i] = k;
            }
            int ss = seg[1 + 2 * ns], se = seg[2 + 2 * ns];
            const int ah = seg[3 + 2 * ns] >> 4, al = seg[3 + 2 * ns] & 15;
            if (!progressive) { ss = 0; se = 63; if (ah || al) return false; }
            else if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || ah > 13 || al > 13) return false;
            JpegBits br{d.data(), d.size(), pos + len};
            for (int i = 0; i < ncomp; ++i) comp[i].pred = 0;
            int eobrun = 0;

            // one 8x8 block of component c in this scan
            auto decode_block = [&](Comp &c, short *blk) -> bool {
                if (!progressive) {
                    const int t = br.decode(hdc[c.td]);
                    if (t < 0 || t > 16) return false;
                    c.pred += jpeg_extend(br.receive(t), t);
                    blk[0] = (short)c.pred;
                    for (int k = 1; k < 64;) {
                        const int rs = br.decode(hac[c.ta]);
                        if (rs < 0) return false;
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r != 15) break;
                            k += 16;
                            continue;
                        }
                        k += r;
                        if (k > 63) return false;
                        blk[zigzag[k]] = (short)jpeg_extend(br.receive(sz), sz);
                        ++k;
                    }
                    return true;
                }
                if (ss == 0) {                                   // DC scan
                    if (ah == 0) {
                        const int t = br.decode(hdc[c.td]);
                        if (t < 0 || t > 16) return false;
                        c.pred += jpeg_extend(br.receive(t), t);
                        blk[0] = (short)(c.pred * (1 << al));
                    } else if (br.bit()) {
                        blk[0] = (short)(blk[0] + (1 << al));
                    }
                    return true;
                }
                if (ah == 0) {                                   // AC, first pass of this band
                    if (eobrun) { --eobrun; return true; }
                    for (int k = ss; k <= se;) {
                        const int rs = br.decode(hac[c.ta]);
                        if (rs < 0) return false;
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r < 15) {
                                eobrun = (1 << r) - 1;
                                if (r) eobrun += br.receive(r);
                                break;
                            }
                            k += 16;
                        } else {
                            k += r;
                            if (k > se) return false;
                            blk[zigzag[k]] = (short)(jpeg_extend(br.receive(sz), sz) * (1 << al));
                            ++k;
                        }
                    }
                    return true;
                }
                // AC refinement (ITU T.81 G.1.2.3): one more bit for every coefficient already non-zero, new
                // coefficients of magnitude 1 << al interleaved with them
                const short bit = (short)(1 << al);
                auto refine = [&](short &p) {
                    if (br.bit() && (p & bit) == 0) p = (short)(p > 0 ? p + bit : p - bit);
                };
                if (eobrun) {
                    --eobrun;
                    for (int k = ss; k <= se; ++k)
                        if (blk[zigzag[k]] != 0) refine(blk[zigzag[k]]);
                    return true;
                }
                int k = ss;
                do {
                    const int rs = br.decode(hac[c.ta]);
                    if (rs < 0) return false;
                    int r = rs >> 4, sv = rs & 15;
                    if (sv == 0) {
                        if (r < 15) {
                            eobrun = (1 << r) - 1;
                            if (r) eobrun += br.receive(r);
                            r = 64;                              // the rest of this block only gets refinement bits
                        }
                    } else {
                        if (sv != 1) return false;
                        sv = br.bit() ? bit : -bit;
                    }
                    while (k <= se) {
                        short &p = blk[zigzag[k++]];
                        if (p != 0) refine(p);
                        else {
                            if (r == 0) { p = (short)sv; break; }
                            --r;
                        }
                    }
                } while (k <= se);
                return true;
            };
            auto at_restart = [&]() {
                br.reset();
                while (br.pos + 1 < br.n && !(br.p[br.pos] == 0xff && br.p[br.pos + 1] >= 0xd0 && br.p[br.pos + 1] <= 0xd7)) ++br.pos;
                br.pos += 2;
                for (int i = 0; i < ncomp; ++i) comp[i].pred = 0;
                eobrun = 0;
            };
            int count = 0;
            if (ns == 1) {
                // a scan of one component covers only the blocks that hold image data, row by row (A.2.3)
                Comp &c = comp[order[0]];
                const int w = ((width * c.h + hmax - 1) / hmax + 7) / 8, h = ((height * c.v + vmax - 1) / vmax + 7) / 8;
                for (int by = 0; by < h; ++by)
                    for (int bx = 0; bx < w; ++bx) {
                        if (restart && count && count % restart == 0) at_restart();
                        ++count;
                        if (!decode_block(c, &c.coef[((size_t)by * c.bw + bx) * 64])) return false;
                    }
            } else {
                for (int my = 0; my < mcuy; ++my)
                    for (int mx = 0; mx < mcux; ++mx) {
                        if (restart && count && count % restart == 0) at_restart();
                        ++count;
                        for (int i = 0; i < ns; ++i) {
                            Comp &c = comp[order[i]];
                            for (int by = 0; by < c.v; ++by)
                                for (int bx = 0; bx < c.h; ++bx)
                                    if (!decode_block(c, &c.coef[((size_t)(my * c.v + by) * c.bw + (size_t)(mx * c.h + bx)) * 64])) return false;
                        }
                    }
            }
            have_scan = true;
            // the next marker: first 0xff followed by something that is neither a stuffed zero nor RSTn
            pos = br.pos;
            while (pos + 1 < d.size() && !(d[pos] == 0xff && d[pos + 1] != 0 && d[pos + 1] != 0xff && !(d[pos + 1] >= 0xd0 && d[pos + 1] <= 0xd7))) ++pos;
            if (!progressive) break;                              // a baseline file is complete after its scan(s) of all components
            continue;
        }
        pos += len;
    }
    if (!width || !have_scan) return false;
    // dequantise (16-bit products, as stb_image keeps them) + inverse DCT into 8-bit planes
    std::vector<unsigned char> plane[4];
    for (int i = 0; i < ncomp; ++i) {
        plane[i].assign((size_t)comp[i].pw * comp[i].ph, 0);
        for (int by = 0; by < comp[i].bh; ++by)
            for (int bx = 0; bx < comp[i].bw; ++bx) {
                const short *cf = &comp[i].coef[((size_t)by * comp[i].bw + bx) * 64];
                short blk[64];
                for (int k = 0; k < 64; ++k) blk[k] = (short)(cf[k] * qt[comp[i].tq][k]);
                jpeg_idct(blk, &plane[i][(size_t)(by * 8) * comp[i].pw + (size_t)bx * 8], comp[i].pw);
            }
    }
    components = ncomp >= 3 ? 3 : 1;                  // (a four-component file - CMYK / YCCK - comes out as RGB)
    rgba.resize((size_t)width * height * 4);
    // rows top to bottom; every component is widened to the full row by stb_image's filters, its two source rows stepping as
    // stb_image steps them (the filter looks up on even output rows, down on odd ones; the last source row repeats)
    struct Up { int hs, vs, ystep, w_lores, ypos, rows; const unsigned char *line0, *line1; std::vector<unsigned char> buf; } up[4];
    for (int i = 0; i < ncomp; ++i) {
        // stb_image takes hmax / h as the integer factor and pads its planes; a file whose factors do not divide the largest
        // one (e.g. H = 4, 3, 1) would be read past the plane here: refused
        if (hmax % comp[i].h != 0 || vmax % comp[i].v != 0) return false;
        up[i].hs = hmax / comp[i].h;
        up[i].vs = vmax / comp[i].v;
        up[i].ystep = up[i].vs >> 1;
        up[i].w_lores = (width + up[i].hs - 1) / up[i].hs;
        up[i].ypos = 0;
        up[i].rows = (height * comp[i].v + vmax - 1) / vmax;
        up[i].line0 = up[i].line1 = plane[i].data();
        up[i].buf.assign((size_t)(up[i].w_lores + 2) * up[i].hs + 8, 0);
    }
    // components that are R, G, B already: ids 'R','G','B', or an Adobe marker with transform 0 and no JFIF marker
    const bool is_rgb = ncomp == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') || (adobe_transform == 0 && !jfif));
    auto fixed = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
    auto mul255 = [](int a, int b) { const unsigned t = (unsigned)(a * b + 128); return (unsigned char)((t + (t >> 8)) >> 8); };      // a * b / 255, rounded
    const int k_cr_r = fixed(1.40200f), k_cr_g = -fixed(0.71414f), k_cb_g = -fixed(0.34414f), k_cb_b = fixed(1.77200f);
    for (int y = 0; y < height; ++y) {
        const unsigned char *row[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < ncomp; ++i) {
            Up &u = up[i];
            const bool lower = u.ystep >= (u.vs >> 1);
            const unsigned char *nr = lower ? u.line1 : u.line0, *fr = lower ? u.line0 : u.line1;
            if (u.hs == 1 && u.vs == 1) row[i] = nr;
            else { jpeg_upsample_row(u.buf.data(), nr, fr, u.w_lores, u.hs, u.vs); row[i] = u.buf.data(); }
            if (++u.ystep >= u.vs) {
                u.ystep = 0;
                u.line0 = u.line1;
                if (++u.ypos < u.rows) u.line1 += comp[i].pw;
            }
        }
        unsigned char *o = &rgba[(size_t)y * width * 4];
        for (int x = 0; x < width; ++x, o += 4) {
            o[3] = 255;
            if (ncomp == 1) { o[0] = o[1] = o[2] = row[0][x]; continue; }
            if (is_rgb) { o[0] = row[0][x]; o[1] = row[1][x]; o[2] = row[2][x]; continue; }
            if (ncomp == 4 && adobe_transform == 0) {            // CMYK as Adobe stores it (inverted): each channel scaled by K
                o[0] = mul255(row[0][x], row[3][x]); o[1] = mul255(row[1][x], row[3][x]); o[2] = mul255(row[2][x], row[3][x]);
                continue;
            }
            // stb_image's YCbCr -> RGB: 20-bit fixed point, the Cb term of green cut to its upper 16 bits
            const int yf = (row[0][x] << 20) + (1 << 19), cb = row[1][x] - 128, cr = row[2][x] - 128;
            const int r = (yf + cr * k_cr_r) >> 20;
            const int g = (int)((unsigned)(yf + cr * k_cr_g) + ((unsigned)(cb * k_cb_g) & 0xffff0000u)) >> 20;
            const int b = (yf + cb * k_cb_b) >> 20;
            o[0] = jclamp(r); o[1] = jclamp(g); o[2] = jclamp(b);
            if (ncomp == 4 && adobe_transform == 2) {            // YCCK: the converted colour is inverted and scaled by K
                o[0] = mul255(255 - o[0], row[3][x]); o[1] = mul255(255 - o[1], row[3][x]); o[2] = mul255(255 - o[2], row[3][x]);
            }                                                    // (four components without an Adobe marker: the fourth is ignored)
        }
    }
    return true;
}

// ---- BMP and TGA: the two other texture formats of stb_image (the reference's reader) that scenes in the wild use.  Both readers
// follow stb_image 2.19's reading of the formats where the formats leave room (tests/test_imageio_reference.py compares with stb_image
// itself): which headers and bit fields are understood, how 5-bit channels are widened, what a missing alpha channel becomes, which
// files are refused.  A read past the end of the file yields zeros, as stb_image's does.
namespace {
struct ByteReader {
    const std::vector<unsigned char> &d;
    size_t pos = 0;
    explicit ByteReader(const std::vector<unsigned char> &v) : d(v) {}
    int u8() { return pos < d.size() ? d[pos++] : (++pos, 0); }
    int u16() { const int lo = u8(); return lo | u8() << 8; }
    uint32_t u32() { const uint32_t lo = (uint32_t)u16(); return lo | (uint32_t)u16() << 16; }
    void skip(long n) { if (n > 0) pos += (size_t)n; }              // (stb_image ignores negative skips too)
};
int high_bit(uint32_t z) { int n = -1; while (z) { ++n; z >>= 1; } return n; }
int bit_count(uint32_t z) { int n = 0; for (; z; z &= z - 1) ++n; return n; }
// a channel of `bits` bits under a mask, moved to bit 7 and widened to 8 bits by replicating it (stb_image's table: x * 0xff, 0x55,
// 0x49 >> 1, 0x11, 0x21 >> 2, 0x41 >> 4, 0x81 >> 6, 1)
int widen_channel(uint32_t v, int shift, int bits)
{
    static const unsigned mul[9] = {0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01}, shr[9] = {0, 0, 0, 1, 0, 2, 4, 6, 0};
    if (bits < 0 || bits > 8) return 0;
    uint32_t x = shift < 0 ? v << -shift : v >> shift;
    x = (x & 255u) >> (8 - bits);
    return (int)((x * mul[bits]) >> shr[bits]);
}
}  // namespace

bool read_bmp(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba)
{
    std::vector<unsigned char> d;
    if (!read_file(path, d) || d.size() < 26 || d[0] != 'B' || d[1] != 'M') return false;
    ByteReader in(d);
    in.skip(10);
    const long offset = (long)in.u32();
    const long hsz = (long)in.u32();
    if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) return false;
    int64_t w, h;
    if (hsz == 12) { w = in.u16(); h = in.u16(); }
    else { w = (int32_t)in.u32(); h = (int32_t)in.u32(); }
    if (in.u16() != 1) return false;                                 // planes
    const int bpp = in.u16();
    uint32_t mr = 0, mg = 0, mb = 0, ma = 0, all_a = 255;
    if (hsz != 12) {
        const uint32_t compress = in.u32();
        if (compress == 1 || compress == 2) return false;            // RLE
        in.skip(20);
        if (hsz == 40 || hsz == 56) {
            if (hsz == 56) in.skip(16);
            if (bpp == 16 || bpp == 32) {
                if (compress == 0) {
                    if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; all_a = 0; }
                    else { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
                } else if (compress == 3) {
                    mr = in.u32(); mg = in.u32(); mb = in.u32();
                    if (mr == mg && mg == mb) return false;
                } else return false;
            }
        } else {
            mr = in.u32(); mg = in.u32(); mb = in.u32(); ma = in.u32();
            in.skip(4 + 48);
            if (hsz == 124) in.skip(16);
        }
    }
    const bool bottom_up = h > 0;
    if (h < 0) h = -h;
    if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24) || w * h > ((int64_t)1 << 28)) return false;
    if ((uint64_t)w * (uint64_t)h * (uint64_t)bpp / 8 > (uint64_t)d.size()) return false;      // more pixels than the file holds
    width = (int)w;
    height = (int)h;
    long psize = 0;
    if (hsz == 12) { if (bpp < 24) psize = (offset - 14 - 24) / 3; }
    else if (bpp < 16) psize = (offset - 14 - hsz) >> 2;
    components = ma ? 4 : 3;
    rgba.assign((size_t)width * height * 4, 255);
    auto row_of = [&](int j) { return &rgba[(size_t)(bottom_up ? height - 1 - j : j) * width * 4]; };
    if (bpp < 16) {
        if (psize <= 0 || psize > 256) return false;
        unsigned char pal[256][3] = {{0}};
        for (long i = 0; i < psize; ++i) {
            pal[i][2] = (unsigned char)in.u8(); pal[i][1] = (unsigned char)in.u8(); pal[i][0] = (unsigned char)in.u8();
            if (hsz != 12) in.u8();
        }
        in.skip(offset - 14 - hsz - psize * (hsz == 12 ? 3 : 4));
        if (bpp != 1 && bpp != 4 && bpp != 8) return false;
        const int row_bytes = bpp == 1 ? (width + 7) >> 3 : bpp == 4 ? (width + 1) >> 1 : width;
        const int pad = (-row_bytes) & 3;
        for (int j = 0; j < height; ++j) {
            unsigned char *o = row_of(j);
            int v = 0;
            for (int i = 0; i < width; ++i) {
                int k;
                if (bpp == 8) k = in.u8();
                else if (bpp == 4) { if (!(i & 1)) v = in.u8(); k = (i & 1) ? v & 15 : v >> 4; }
                else { if (!(i & 7)) v = in.u8(); k = (v >> (7 - (i & 7))) & 1; }
                o[4 * i] = pal[k][0]; o[4 * i + 1] = pal[k][1]; o[4 * i + 2] = pal[k][2];
            }
            if (bpp == 1 && !(width & 7)) in.u8();                  // (stb_image fetches the next byte after every eighth pixel, also the last)
            in.skip(pad);
        }
    } else {
        in.skip(offset - 14 - hsz);
        const int row_bytes = bpp == 24 ? 3 * width : bpp == 16 ? 2 * width : 0;
        const int pad = (-row_bytes) & 3;
        int easy = 0;
        if (bpp == 24) easy = 1;
        else if (bpp == 32 && mb == 0xffu && mg == 0xff00u && mr == 0x00ff0000u && ma == 0xff000000u) easy = 2;
        else if (bpp != 16 && bpp != 32) return false;
        int rs = 0, gs = 0, bs = 0, as = 0, rc = 0, gc = 0, bc = 0, ac = 0;
        if (!easy) {
            if (!mr || !mg || !mb) return false;
            rs = high_bit(mr) - 7; rc = bit_count(mr);
            gs = high_bit(mg) - 7; gc = bit_count(mg);
            bs = high_bit(mb) - 7; bc = bit_count(mb);
            as = high_bit(ma) - 7; ac = bit_count(ma);
        }
        for (int j = 0; j < height; ++j) {
            unsigned char *o = row_of(j);
            for (int i = 0; i < width; ++i, o += 4) {
                unsigned a;
                if (easy) {
                    o[2] = (unsigned char)in.u8(); o[1] = (unsigned char)in.u8(); o[0] = (unsigned char)in.u8();
                    a = easy == 2 ? (unsigned)in.u8() : 255u;
                } else {
                    const uint32_t v = bpp == 16 ? (uint32_t)in.u16() : in.u32();
                    o[0] = (unsigned char)widen_channel(v & mr, rs, rc);
                    o[1] = (unsigned char)widen_channel(v & mg, gs, gc);
                    o[2] = (unsigned char)widen_channel(v & mb, bs, bc);
                    a = ma ? (unsigned)widen_channel(v & ma, as, ac) : 255u;
                }
                all_a |= a;
                o[3] = (unsigned char)a;
            }
            in.skip(pad);
        }
    }
    if (components == 4 && all_a == 0)                               // an alpha channel that is zero everywhere is not one
        for (size_t i = 3; i < rgba.size(); i += 4) rgba[i] = 255;
    if (components == 3)
        for (size_t i = 3; i < rgba.size(); i += 4) rgba[i] = 255;
    return true;
}

bool read_tga(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba)
{
    std::vector<unsigned char> d;
    if (!read_file(path, d) || d.size() < 18) return false;
    ByteReader in(d);
    const int id_len = in.u8(), colour_map = in.u8();
    int type = in.u8();
    const int pal_start = in.u16(), pal_len = in.u16(), pal_bits = in.u8();
    in.skip(4);                                                     // x, y origin
    const int w = in.u16(), h = in.u16(), bpp = in.u8(), descriptor = in.u8();
    // what stb_image recognises as a TGA at all (the format has no signature)
    if (colour_map > 1) return false;
    if (colour_map == 1) {
        if (type != 1 && type != 9) return false;
        if (pal_bits != 8 && pal_bits != 15 && pal_bits != 16 && pal_bits != 24 && pal_bits != 32) return false;
        if (bpp != 8 && bpp != 16) return false;
    } else if (type != 2 && type != 3 && type != 10 && type != 11) return false;
    if (w < 1 || h < 1) return false;
    if (bpp != 8 && bpp != 15 && bpp != 16 && bpp != 24 && bpp != 32) return false;
    const bool rle = type >= 8;
    if (rle) type -= 8;
    const bool top_down = (descriptor >> 5) & 1;
    // channels: from the palette entries of a colour-mapped file, else from the pixel size; 15 / 16 bits are 5-5-5 colour
    // unless the file is grey (then 16 bits are grey + alpha)
    auto channels = [](int bits, bool grey, bool &rgb16) {
        rgb16 = false;
        switch (bits) {
        case 8: return 1;
        case 16: if (grey) return 2;  /* fall through */
        case 15: rgb16 = true; return 3;
        case 24: return 3;
        case 32: return 4;
        default: return 0;
        }
    };
    bool rgb16 = false;
    const int comp = colour_map ? channels(pal_bits, false, rgb16) : channels(bpp, type == 3, rgb16);
    if (!comp) return false;
    if ((int64_t)w * h > ((int64_t)1 << 28)) return false;
    // a header must not claim more pixels than the file can hold (a run-length packet yields at most 128 pixels from 2 bytes):
    // stb_image hands back uninitialised rows for such files, here they are refused
    const uint64_t px_bytes = (uint64_t)((bpp + 7) / 8);
    if (rle ? (uint64_t)w * h > 128u * (uint64_t)d.size() : (uint64_t)w * h * px_bytes > (uint64_t)d.size()) return false;
    width = w; height = h; components = comp;
    in.skip(id_len);
    auto read555 = [&](unsigned char *o) {
        const int px = in.u16();
        o[0] = (unsigned char)((((px >> 10) & 31) * 255) / 31);
        o[1] = (unsigned char)((((px >> 5) & 31) * 255) / 31);
        o[2] = (unsigned char)(((px & 31) * 255) / 31);
    };
    std::vector<unsigned char> palette;
    if (colour_map) {
        if (pal_len == 0) return false;                             // (stb_image reads entry 0 of an empty table)
        in.skip(pal_start);
        palette.assign((size_t)pal_len * comp, 0);
        if (rgb16) for (int i = 0; i < pal_len; ++i) read555(&palette[(size_t)i * comp]);
        else {
            if (in.pos + palette.size() > d.size()) return false;    // "bad palette"
            for (unsigned char &b : palette) b = (unsigned char)in.u8();
        }
    }
    std::vector<unsigned char> px((size_t)w * h * comp);
    unsigned char raw[4] = {0, 0, 0, 0};
    int run = 0;
    bool repeating = false;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        bool fetch = true;
        if (rle) {
            if (run == 0) { const int cmd = in.u8(); run = 1 + (cmd & 127); repeating = cmd >> 7; }
            else if (repeating) fetch = false;
        }
        if (fetch) {
            if (colour_map) {
                int k = bpp == 8 ? in.u8() : in.u16();
                if (k >= pal_len) k = 0;
                for (int j = 0; j < comp; ++j) raw[j] = palette[(size_t)k * comp + j];
            } else if (rgb16) read555(raw);
            else for (int j = 0; j < comp; ++j) raw[j] = (unsigned char)in.u8();
        }
        for (int j = 0; j < comp; ++j) px[i * comp + j] = raw[j];
        --run;
    }
    rgba.resize((size_t)w * h * 4);
    for (int y = 0; y < h; ++y) {
        const unsigned char *src = &px[(size_t)(top_down ? y : h - 1 - y) * w * comp];
        unsigned char *o = &rgba[(size_t)y * w * 4];
        for (int x = 0; x < w; ++x, src += comp, o += 4) {
            if (comp == 1) { o[0] = o[1] = o[2] = src[0]; o[3] = 255; }
            else if (comp == 2) { o[0] = o[1] = o[2] = src[0]; o[3] = src[1]; }
            else if (rgb16) { o[0] = src[0]; o[1] = src[1]; o[2] = src[2]; o[3] = 255; }
            else { o[0] = src[2]; o[1] = src[1]; o[2] = src[0]; o[3] = comp == 4 ? src[3] : 255; }     // stored B, G, R
        }
    }
    return true;
}

// binary PGM / PPM (P5 / P6) as stb_image reads them: 8-bit samples taken as they are (not scaled by the maximum value), '#' comments
// between the header fields, the pixels starting right after the single character that ends the maximum value
bool read_pnm(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba)
{
    std::vector<unsigned char> d;
    if (!read_file(path, d) || d.size() < 3 || d[0] != 'P' || (d[1] != '5' && d[1] != '6')) return false;
    ByteReader in(d);
    in.skip(2);
    const int comp = d[1] == '6' ? 3 : 1;
    int c = in.u8();
    auto at_end = [&]() { return in.pos >= d.size(); };
    auto blanks = [&]() {
        for (;;) {
            while (!at_end() && (c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r')) c = in.u8();
            if (at_end() || c != '#') break;
            while (!at_end() && c != '\n' && c != '\r') c = in.u8();
        }
    };
    auto number = [&]() { int64_t v = 0; while (!at_end() && c >= '0' && c <= '9') { v = v * 10 + (c - '0'); if (v > (1 << 28)) v = 1 << 28; c = in.u8(); } return (int)v; };
    blanks();
    const int w = number();
    blanks();
    const int h = number();
    blanks();
    const int maxv = number();
    if (maxv > 255 || w <= 0 || h <= 0) return false;
    if ((uint64_t)w * (uint64_t)h * (uint64_t)comp > (uint64_t)d.size()) return false;      // more pixels than the file holds
    width = w; height = h; components = comp;
    rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        unsigned char *o = &rgba[4 * i];
        if (comp == 1) o[0] = o[1] = o[2] = (unsigned char)in.u8();
        else { o[0] = (unsigned char)in.u8(); o[1] = (unsigned char)in.u8(); o[2] = (unsigned char)in.u8(); }
        o[3] = 255;
    }
    return true;
}

// the formats in the order stb_image tries them (TGA last: it has no signature)
bool read_any8(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba)
{
    return read_jpeg(path, width, height, components, rgba) || read_png(path, width, height, components, rgba) ||
           read_bmp(path, width, height, components, rgba) || read_pnm(path, width, height, components, rgba) ||
           read_tga(path, width, height, components, rgba);
}

// ImageIO::LoadTexture + Texture::Texture (src/imageio.cpp:11-59, src/texture.h:15-27):
// flip vertically, 1/255, sRGB -> linear by powf(x, 2.2f) on r,g,b, then truncate x*255 back to uchar.
bool load_texture(const char *path, int &width, int &height, std::vector<gpt_uchar4> &texels)
{
    std::vector<unsigned char> rgba;
    int comp = 0;
    if (!read_any8(path, width, height, comp, rgba)) return false;
    texels.resize((size_t)width * height);
    const float inv = 1.f / 255.f;
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const unsigned char *p = &rgba[((size_t)(height - 1 - y) * width + x) * 4];   // stbi flip-on-load
            float r = p[0] * inv, g = p[1] * inv, b = p[2] * inv, a = comp == 4 || comp == 2 ? p[3] * inv : 1.f;
            r = std::pow(r, 2.2f); g = std::pow(g, 2.2f); b = std::pow(b, 2.2f);
            gpt_uchar4 t;
            t.x = (unsigned char)(r * 255); t.y = (unsigned char)(g * 255); t.z = (unsigned char)(b * 255);
            t.w = (unsigned char)(a * 255);
            texels[(size_t)y * width + x] = t;
        }
    }
    return true;
}

bool write_pfm(const char *path, int width, int height, const float *rgb_bottom_up)
{
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
    size_t n = (size_t)width * height * 3;
    size_t w = std::fwrite(rgb_bottom_up, sizeof(float), n, f);
    std::fclose(f);
    return w == n;
}

// PFM -> float3 rows TOP-DOWN (the orientation tinyexr hands the reference: row 0 = top of the image)
bool read_pfm_top_down(const char *path, int &width, int &height, std::vector<gpt_float3> &out)
{
    std::vector<unsigned char> d;
    if (!read_file(path, d) || d.size() < 8) return false;
    int ch = d[0] == 'P' && d[1] == 'F' ? 3 : (d[0] == 'P' && d[1] == 'f' ? 1 : 0);
    if (!ch) return false;
    size_t pos = 2;
    double vals[3];
    for (int k = 0; k < 3; ++k) {
        while (pos < d.size() && (d[pos] == ' ' || d[pos] == '\n' || d[pos] == '\r' || d[pos] == '\t')) ++pos;
        char buf[64];
        size_t l = 0;
        while (pos < d.size() && l < 63 && !(d[pos] == ' ' || d[pos] == '\n' || d[pos] == '\r' || d[pos] == '\t')) buf[l++] = (char)d[pos++];
        buf[l] = 0;
        vals[k] = std::atof(buf);
    }
    ++pos;   // the single whitespace after the scale
    width = (int)vals[0]; height = (int)vals[1];
    const bool little = vals[2] < 0;
    if (width <= 0 || height <= 0 || pos + (size_t)width * height * ch * 4 > d.size()) return false;
    out.resize((size_t)width * height);
    auto rd = [&](size_t idx) {
        unsigned char b[4];
        std::memcpy(b, &d[pos + idx * 4], 4);
        if (!little) { unsigned char t = b[0]; b[0] = b[3]; b[3] = t; t = b[1]; b[1] = b[2]; b[2] = t; }
        float f;
        std::memcpy(&f, b, 4);
        return f;
    };
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            size_t src = ((size_t)(height - 1 - y) * width + x) * ch;
            gpt_float3 c;
            if (ch == 3) { c.x = rd(src); c.y = rd(src + 1); c.z = rd(src + 2); }
            else { c.x = c.y = c.z = rd(src); }
            out[(size_t)y * width + x] = c;
        }
    return true;
}

// ---- OpenEXR, scanline images (what ImageIO::LoadExr hands the reference through tinyexr's LoadEXR:
// float RGB, row 0 = top).  Supported: single-part scanline files, compression NONE / RLE / ZIPS / ZIP / PIZ,
// HALF / FLOAT / UINT channels named R,G,B (or a single luminance channel Y), sampling 1x1.  Not supported
// (returns false): tiled or multi-part files, PXR24 / B44 / DWA compression (tinyexr, the reference's reader, has none of those either).
namespace {
float half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400u));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
    else bits = sign | (exp + 112u) << 23 | man << 13;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}
bool exr_rle_decode(const unsigned char *in, size_t n, std::vector<unsigned char> &out, size_t expect)
{
    out.clear();
    size_t i = 0;
    while (i < n) {
        int c = (signed char)in[i++];
        if (c < 0) {
            size_t cnt = (size_t)(-c);
            if (i + cnt > n) return false;
            out.insert(out.end(), in + i, in + i + cnt);
            i += cnt;
        } else {
            if (i >= n) return false;
            out.insert(out.end(), (size_t)c + 1, in[i++]);
        }
        if (out.size() > expect) return false;
    }
    return out.size() == expect;
}
void exr_unpredict(std::vector<unsigned char> &buf, std::vector<unsigned char> &tmp)
{
    for (size_t i = 1; i < buf.size(); ++i) buf[i] = (unsigned char)(buf[i - 1] + buf[i] - 128);
    tmp.resize(buf.size());
    const size_t half = (buf.size() + 1) / 2;
    for (size_t i = 0, a = 0, b = half; i < buf.size();) {
        tmp[i++] = buf[a++];
        if (i < buf.size()) tmp[i++] = buf[b++];
    }
    buf.swap(tmp);
}

// ---- PIZ (OpenEXR's wavelet + Huffman scheme, as published in the OpenEXR sources' ImfPizCompressor / ImfHuf / ImfWav and
// decoded by tinyexr for the reference).  A block of up to 32 scanlines is stored as: the range of 16-bit values in use
// (a bitmap), then a Huffman-coded stream of 16-bit symbols holding, channel after channel, the 2-D Haar-like wavelet
// transform of the block's values mapped through that bitmap.
struct PizBits {                                    // most-significant-bit-first reader
    const unsigned char *p, *end;
    uint64_t acc = 0;
    int n = 0;
    bool need(int k) { while (n < k) { if (p >= end) return false; acc = acc << 8 | *p++; n += 8; } return true; }
    uint32_t take(int k) { n -= k; return (uint32_t)(acc >> n) & ((1u << k) - 1u); }
};
constexpr int kHufEncSize = 65537, kHufDecBits = 14, kHufDecSize = 1 << kHufDecBits;
struct HufDec { int len = 0; int lit = 0; std::vector<int> longs; };   // len > 0: short code of `lit`; else: the long codes under this prefix

// code lengths im..iM (6 bits each, with zero runs) -> canonical codes: table[s] = length | code << 6
bool huf_read_code_table(const unsigned char *src, size_t n, int im, int iM, std::vector<uint64_t> &table, size_t &used)
{
    PizBits b{src, src + n};
    table.assign(kHufEncSize, 0);
    for (int s = im; s <= iM; ++s) {
        if (!b.need(6)) return false;
        const uint32_t l = b.take(6);
        if (l == 63) {                              // a long run of unused symbols
            if (!b.need(8)) return false;
            const int run = (int)b.take(8) + 6;
            if (s + run > iM + 1) return false;
            s += run - 1;
        } else if (l >= 59) {                       // a short run
            const int run = (int)l - 59 + 2;
            if (s + run > iM + 1) return false;
            s += run - 1;
        } else table[(size_t)s] = l;
    }
    used = (size_t)(b.p - src);
    uint64_t first[59] = {0}, c = 0;
    for (int s = 0; s < kHufEncSize; ++s) first[table[(size_t)s]] += 1;
    for (int l = 58; l > 0; --l) { const uint64_t next = (c + first[l]) >> 1; first[l] = c; c = next; }
    for (int s = 0; s < kHufEncSize; ++s) {
        const uint64_t l = table[(size_t)s];
        if (l > 0) table[(size_t)s] = l | first[l]++ << 6;
    }
    return true;
}

bool huf_build_decoder(const std::vector<uint64_t> &table, int im, int iM, std::vector<HufDec> &dec)
{
    dec.assign(kHufDecSize, HufDec());
    for (int s = im; s <= iM; ++s) {
        const uint64_t c = table[(size_t)s] >> 6;
        const int l = (int)(table[(size_t)s] & 63);
        if (l == 0) continue;
        if (c >> l) return false;
        if (l > kHufDecBits) {
            HufDec &d = dec[(size_t)(c >> (l - kHufDecBits))];
            if (d.len) return false;
            d.longs.push_back(s);
        } else {
            const size_t lo = (size_t)(c << (kHufDecBits - l)), cnt = (size_t)1 << (kHufDecBits - l);
            for (size_t i = 0; i < cnt; ++i) {
                HufDec &d = dec[lo + i];
                if (d.len || !d.longs.empty()) return false;
                d.len = l;
                d.lit = s;
            }
        }
    }
    return true;
}

bool huf_decode(const std::vector<uint64_t> &table, const std::vector<HufDec> &dec, const unsigned char *src, size_t n_bits, int rlc,
                std::vector<uint16_t> &out, size_t n_out)
{
    const unsigned char *in = src, *ie = src + (n_bits + 7) / 8;
    uint64_t c = 0;
    int lc = 0;
    out.clear();
    out.reserve(n_out);
    auto emit = [&](int sym) -> bool {             // a symbol, or (the run-length symbol) a repeat of the last one
        if (sym == rlc) {
            if (lc < 8) { if (in >= ie) return false; c = c << 8 | *in++; lc += 8; }
            lc -= 8;
            const size_t cnt = (size_t)(c >> lc) & 0xff;
            if (out.empty() || out.size() + cnt > n_out) return false;
            out.insert(out.end(), cnt, out.back());
        } else {
            if (out.size() >= n_out) return false;
            out.push_back((uint16_t)sym);
        }
        return true;
    };
    auto long_code = [&](const HufDec &d) -> bool {
        for (int s : d.longs) {
            const int l = (int)(table[(size_t)s] & 63);
            while (lc < l && in < ie) { c = c << 8 | *in++; lc += 8; }
            if (lc >= l && (table[(size_t)s] >> 6) == ((c >> (lc - l)) & (((uint64_t)1 << l) - 1))) { lc -= l; return emit(s); }
        }
        return false;
    };
    while (in < ie) {
        c = c << 8 | *in++;
        lc += 8;
        while (lc >= kHufDecBits) {
            const HufDec &d = dec[(size_t)(c >> (lc - kHufDecBits)) & (kHufDecSize - 1)];
            if (d.len) { lc -= d.len; if (!emit(d.lit)) return false; }
            else if (d.longs.empty() || !long_code(d)) return false;
        }
    }
    const int pad = (int)((8 - n_bits) & 7);
    c >>= pad;
    lc -= pad;
    while (lc > 0) {
        const HufDec &d = dec[(size_t)(c << (kHufDecBits - lc)) & (kHufDecSize - 1)];
        if (!d.len || d.len > lc) return false;
        lc -= d.len;
        if (!emit(d.lit)) return false;
    }
    return out.size() == n_out;
}

bool huf_uncompress(const unsigned char *src, size_t n, std::vector<uint16_t> &out, size_t n_out)
{
    if (n == 0) return n_out == 0;
    if (n < 20) return false;
    auto u32 = [&](size_t p) { return (uint32_t)src[p] | (uint32_t)src[p + 1] << 8 | (uint32_t)src[p + 2] << 16 | (uint32_t)src[p + 3] << 24; };
    const uint32_t im = u32(0), iM = u32(4), n_bits = u32(12);
    if (im >= (uint32_t)kHufEncSize || iM >= (uint32_t)kHufEncSize || im > iM) return false;
    std::vector<uint64_t> table;
    size_t used = 0;
    if (!huf_read_code_table(src + 20, n - 20, (int)im, (int)iM, table, used)) return false;
    if (((uint64_t)n_bits + 7) / 8 > (uint64_t)(n - 20 - used)) return false;      // (in 64 bits: n_bits + 7 wraps in 32)
    std::vector<HufDec> dec;
    if (!huf_build_decoder(table, (int)im, (int)iM, dec)) return false;
    return huf_decode(table, dec, src + 20 + used, n_bits, (int)iM, out, n_out);
}

// The inverse of PIZ's 2-D Haar-style wavelet over an nx * ny grid of 16-bit values (strides ox, oy in elements); mx: the
// largest value of the data.  Level by level from the coarsest: the grid points of level p are the multiples of p in both
// directions (nx / p columns, ny / p rows), and a level undoes first the vertical lifting pairs {row 2j p, row (2j + 1) p}
// in every column of the level, then the horizontal pairs {column 2j p, column (2j + 1) p} in every row of the level - an
// odd last row / column of a level has no partner in that direction and is left as it is.  (The encoder's forward transform
// pairs columns first and rows second within 2 x 2 blocks; block by block or pass by pass, every value sees the same two
// operations in the same order.)  A pair (low, high) becomes the two samples it was made from: for data below 2^14 by
// average / difference in signed 16-bit arithmetic, otherwise modulo 2^16 with the offset 0x8000.
void wav_decode(uint16_t *v, int nx, int ox, int ny, int oy, uint16_t mx)
{
    const bool small_range = mx < (1 << 14);
    auto unlift = [small_range](uint16_t &low, uint16_t &high) {
        if (small_range) {
            const int l = (int16_t)low, h = (int16_t)high;
            const int first = l + (h & 1) + (h >> 1);
            low = (uint16_t)(int16_t)first;
            high = (uint16_t)(int16_t)(first - h);
        } else {
            const int second = ((int)low - ((int)high >> 1)) & 0xffff;
            const int first = ((int)high + second - 0x8000) & 0xffff;
            low = (uint16_t)first;
            high = (uint16_t)second;
        }
    };
    const int shorter = nx < ny ? nx : ny;
    int coarsest = 1;
    while (2 * coarsest <= shorter) coarsest *= 2;                // the largest power of two that fits the shorter side
    for (int p = coarsest / 2; p >= 1; p /= 2) {
        const int cols = nx / p, rows = ny / p;
        for (int j = 0; j < rows / 2; ++j) {
            uint16_t *upper = v + (ptrdiff_t)oy * (2 * j) * p, *lower = upper + (ptrdiff_t)oy * p;
            for (int k = 0; k < cols; ++k) unlift(upper[(ptrdiff_t)ox * k * p], lower[(ptrdiff_t)ox * k * p]);
        }
        for (int k = 0; k < rows; ++k) {
            uint16_t *row = v + (ptrdiff_t)oy * k * p;
            for (int j = 0; j < cols / 2; ++j) unlift(row[(ptrdiff_t)ox * (2 * j) * p], row[(ptrdiff_t)ox * (2 * j + 1) * p]);
        }
    }
}

// one PIZ block -> `lines` scanlines in the uncompressed layout (per line: channel after channel).  words[c]: 16-bit words per
// value of channel c (1 HALF, 2 FLOAT / UINT)
bool exr_piz_decode(const unsigned char *src, size_t n, int width, int lines, const std::vector<int> &words, std::vector<unsigned char> &raw)
{
    if (n < 4) return false;
    const int lo = src[0] | src[1] << 8, hi = src[2] | src[3] << 8;
    std::vector<unsigned char> bitmap(8192, 0);
    size_t pos = 4;
    if (lo <= hi) {
        if (hi >= 8192 || pos + (size_t)(hi - lo + 1) > n) return false;
        std::memcpy(&bitmap[(size_t)lo], src + pos, (size_t)(hi - lo + 1));
        pos += (size_t)(hi - lo + 1);
    }
    std::vector<uint16_t> lut(65536, 0);
    int k = 0;
    for (int i = 0; i < 65536; ++i)
        if (i == 0 || (bitmap[(size_t)i >> 3] & (1 << (i & 7)))) lut[(size_t)k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    if (pos + 4 > n) return false;
    const uint32_t len = (uint32_t)src[pos] | (uint32_t)src[pos + 1] << 8 | (uint32_t)src[pos + 2] << 16 | (uint32_t)src[pos + 3] << 24;
    pos += 4;
    if (len > n - pos) return false;
    size_t total = 0;
    for (int wds : words) total += (size_t)width * lines * wds;
    std::vector<uint16_t> v;
    if (!huf_uncompress(src + pos, len, v, total)) return false;
    size_t start = 0;
    std::vector<size_t> chan_start(words.size());
    for (size_t c = 0; c < words.size(); ++c) {
        chan_start[c] = start;
        for (int j = 0; j < words[c]; ++j) wav_decode(&v[start + (size_t)j], width, words[c], lines, width * words[c], max_value);
        start += (size_t)width * lines * words[c];
    }
    for (uint16_t &x : v) x = lut[x];
    raw.resize(total * 2);
    unsigned char *o = raw.data();
    for (int y = 0; y < lines; ++y)
        for (size_t c = 0; c < words.size(); ++c) {
            const size_t cnt = (size_t)width * words[c];
            const uint16_t *q = &v[chan_start[c] + (size_t)y * cnt];
            for (size_t i = 0; i < cnt; ++i) { *o++ = (unsigned char)(q[i] & 0xff); *o++ = (unsigned char)(q[i] >> 8); }
        }
    return true;
}
}  // namespace

bool read_exr_rgb_top_down(const char *path, int &width, int &height, std::vector<gpt_float3> &out)
{
    std::vector<unsigned char> d;
    if (!read_file(path, d) || d.size() < 16) return false;
    auto le32 = [&](size_t p) { return (uint32_t)d[p] | (uint32_t)d[p + 1] << 8 | (uint32_t)d[p + 2] << 16 | (uint32_t)d[p + 3] << 24; };
    if (le32(0) != 20000630u) return false;
    const uint32_t version = le32(4);
    if ((version & 0xff) != 2 || (version & 0x1a00u)) return false;     // tiled / deep / multipart
    size_t pos = 8;
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans;
    int compression = -1, xmin = 0, ymin = 0, xmax = -1, ymax = -1;
    while (pos < d.size() && d[pos] != 0) {
        std::string name, type;
        while (pos < d.size() && d[pos]) name += (char)d[pos++];
        ++pos;
        while (pos < d.size() && d[pos]) type += (char)d[pos++];
        ++pos;
        if (pos + 4 > d.size()) return false;
        const uint32_t size = le32(pos);
        pos += 4;
        if (pos + size > d.size()) return false;
        if (name == "channels") {
            size_t q = pos;
            while (q < pos + size && d[q] != 0) {
                Chan c;
                while (q < d.size() && d[q]) c.name += (char)d[q++];
                ++q;
                if (q + 16 > d.size()) return false;
                c.type = (int)le32(q);
                const uint32_t xs = le32(q + 8), ys = le32(q + 12);
                if (xs != 1 || ys != 1) return false;
                q += 16;
                chans.push_back(c);
            }
        } else if (name == "compression") compression = d[pos];
        else if (name == "dataWindow" && size == 16) {
            xmin = (int)le32(pos); ymin = (int)le32(pos + 4); xmax = (int)le32(pos + 8); ymax = (int)le32(pos + 12);
        }
        pos += size;
    }
    ++pos;
    const int64_t w64 = (int64_t)xmax - (int64_t)xmin + 1, h64 = (int64_t)ymax - (int64_t)ymin + 1;
    if (w64 <= 0 || h64 <= 0 || w64 > (1 << 20) || h64 > (1 << 20) || w64 * h64 > ((int64_t)1 << 28)) return false;
    width = (int)w64;
    height = (int)h64;
    if (chans.empty() || compression < 0 || compression > 4) return false;
    const int lines_per_block = compression == 4 ? 32 : compression == 3 ? 16 : 1;
    const int n_blocks = (height + lines_per_block - 1) / lines_per_block;
    if (pos + (size_t)n_blocks * 8 > d.size()) return false;
    std::vector<size_t> choff(chans.size());
    std::vector<int> words;
    for (const Chan &c : chans) words.push_back(c.type == 1 ? 1 : 2);
    size_t line_bytes = 0;
    int ir = -1, ig = -1, ib = -1, iy = -1;
    for (size_t c = 0; c < chans.size(); ++c) {
        choff[c] = line_bytes;
        line_bytes += (size_t)width * (chans[c].type == 1 ? 2 : 4);
        if (chans[c].name == "R") ir = (int)c;
        else if (chans[c].name == "G") ig = (int)c;
        else if (chans[c].name == "B") ib = (int)c;
        else if (chans[c].name == "Y") iy = (int)c;
    }
    if ((ir < 0 || ig < 0 || ib < 0) && iy < 0) return false;
    out.assign((size_t)width * height, gpt_float3{0, 0, 0});
    std::vector<unsigned char> raw, tmp;
    auto sample = [&](const unsigned char *line, int c, int x) -> float {
        const unsigned char *p = line + choff[(size_t)c];
        if (chans[(size_t)c].type == 1) { uint16_t h = (uint16_t)(p[2 * x] | p[2 * x + 1] << 8); return half_to_float(h); }
        uint32_t u = (uint32_t)p[4 * x] | (uint32_t)p[4 * x + 1] << 8 | (uint32_t)p[4 * x + 2] << 16 | (uint32_t)p[4 * x + 3] << 24;
        if (chans[(size_t)c].type == 0) return (float)u;
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    for (int b = 0; b < n_blocks; ++b) {
        uint64_t off = 0;
        for (int k = 7; k >= 0; --k) off = off << 8 | d[pos + (size_t)b * 8 + (size_t)k];
        if (d.size() < 8 || off > d.size() - 8) return false;          // (no addition: off is a 64-bit field of the file)
        const int64_t y0_64 = (int64_t)(int32_t)le32((size_t)off) - (int64_t)ymin;
        const uint32_t csize = le32((size_t)off + 4);
        if (csize > d.size() - 8 - (size_t)off || y0_64 < 0 || y0_64 >= height) return false;
        const int y0 = (int)y0_64;
        const int lines = y0 + lines_per_block <= height ? lines_per_block : height - y0;
        const size_t expect = line_bytes * (size_t)lines;
        const unsigned char *src = &d[(size_t)off + 8];
        if (compression == 0 || csize == expect) raw.assign(src, src + csize);
        else if (compression == 1) {
            if (!exr_rle_decode(src, csize, raw, expect)) return false;
            exr_unpredict(raw, tmp);
        } else if (compression == 4) {
            if (!exr_piz_decode(src, csize, width, lines, words, raw)) return false;
        } else {
            raw.clear();
            if (csize < 6 || !inflate_raw(src + 2, csize - 2, raw, expect)) return false;
            exr_unpredict(raw, tmp);
        }
        if (raw.size() < expect) return false;
        for (int l = 0; l < lines; ++l) {
            const unsigned char *line = raw.data() + line_bytes * (size_t)l;
            gpt_float3 *row = &out[(size_t)(y0 + l) * width];
            for (int x = 0; x < width; ++x) {
                if (ir >= 0 && ig >= 0 && ib >= 0) row[x] = gpt_float3{sample(line, ir, x), sample(line, ig, x), sample(line, ib, x)};
                else { float v = sample(line, iy, x); row[x] = gpt_float3{v, v, v}; }
            }
        }
    }
    return true;
}

}  // namespace imageio

extern "C" {

int gpt_save_png(const char *path, int32_t width, int32_t height, const float *rgb)
{
    if (!path || !rgb || width <= 0 || height <= 0) { gpt_set_error("gpt_save_png: invalid argument"); return GPT_ERR_INVALID_ARG; }
    std::vector<unsigned char> px((size_t)width * height * 3);
    for (int i = 0; i < height; ++i)
        for (int j = 0; j < width; ++j) {
            const size_t pixel = (size_t)i * width + j, inverse = (size_t)(height - i - 1) * width + j;   // imageio.cpp:64-66
            for (int c = 0; c < 3; ++c) {
                float v = rgb[3 * inverse + c];
                v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
                if (!(v == v)) v = 0.f;
                px[3 * pixel + c] = (unsigned char)(unsigned)(v * 255.f);
            }
        }
    if (!imageio::write_png_rgb8(path, width, height, px.data())) { gpt_set_error("gpt_save_png: cannot write %s", path); return GPT_ERR_IO; }
    return GPT_OK;
}

// ImageIO::SaveExr (src/imageio.cpp:104-161): the reference stores B, G, R as HALF through tinyexr.  Same
// channels and pixel type here, scanline, uncompressed; rows are written top-down from the bottom-up film.
static uint16_t float_to_half(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t man = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
    if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - exp;
        uint32_t h = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (h & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = (uint32_t)exp << 10 | man >> 13;
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;      // round to nearest even; may carry into the exponent
    return (uint16_t)(sign | h);
}

int gpt_save_exr(const char *path, int32_t width, int32_t height, const float *rgb)
{
    if (!path || !rgb || width <= 0 || height <= 0) { gpt_set_error("gpt_save_exr: invalid argument"); return GPT_ERR_INVALID_ARG; }
    std::vector<unsigned char> out;
    auto put32 = [&](uint32_t v) { for (int k = 0; k < 4; ++k) out.push_back((unsigned char)(v >> (8 * k))); };
    auto puts = [&](const char *s) { while (*s) out.push_back((unsigned char)*s++); out.push_back(0); };
    auto attr = [&](const char *name, const char *type, const std::vector<unsigned char> &v) {
        puts(name); puts(type); put32((uint32_t)v.size()); out.insert(out.end(), v.begin(), v.end());
    };
    auto le = [](std::initializer_list<uint32_t> vals) {
        std::vector<unsigned char> v;
        for (uint32_t x : vals) for (int k = 0; k < 4; ++k) v.push_back((unsigned char)(x >> (8 * k)));
        return v;
    };
    put32(20000630u);
    put32(2u);
    std::vector<unsigned char> ch;
    for (const char *n : {"B", "G", "R"}) {
        ch.push_back((unsigned char)n[0]); ch.push_back(0);
        for (uint32_t x : {1u, 0u, 1u, 1u}) for (int k = 0; k < 4; ++k) ch.push_back((unsigned char)(x >> (8 * k)));   // HALF, pLinear+reserved, 1x1
    }
    ch.push_back(0);
    attr("channels", "chlist", ch);
    attr("compression", "compression", {0});
    attr("dataWindow", "box2i", le({0u, 0u, (uint32_t)(width - 1), (uint32_t)(height - 1)}));
    attr("displayWindow", "box2i", le({0u, 0u, (uint32_t)(width - 1), (uint32_t)(height - 1)}));
    attr("lineOrder", "lineOrder", {0});
    const float one = 1.f, zero = 0.f;
    uint32_t one_bits, zero_bits;
    std::memcpy(&one_bits, &one, 4);
    std::memcpy(&zero_bits, &zero, 4);
    attr("pixelAspectRatio", "float", le({one_bits}));
    attr("screenWindowCenter", "v2f", le({zero_bits, zero_bits}));
    attr("screenWindowWidth", "float", le({one_bits}));
    out.push_back(0);
    const size_t line_bytes = (size_t)width * 3 * 2, chunk = 8 + line_bytes;
    const size_t table = out.size();
    for (int y = 0; y < height; ++y) {
        const uint64_t off = table + (size_t)height * 8 + (size_t)y * chunk;
        for (int k = 0; k < 8; ++k) out.push_back((unsigned char)(off >> (8 * k)));
    }
    for (int y = 0; y < height; ++y) {
        put32((uint32_t)y);
        put32((uint32_t)line_bytes);
        const float *row = rgb + (size_t)(height - 1 - y) * width * 3;      // film row 0 = bottom
        for (int c = 2; c >= 0; --c)                                          // B, G, R
            for (int x = 0; x < width; ++x) {
                const uint16_t h = float_to_half(row[3 * x + c]);
                out.push_back((unsigned char)h); out.push_back((unsigned char)(h >> 8));
            }
    }
    FILE *f = std::fopen(path, "wb");
    if (!f) { gpt_set_error("gpt_save_exr: cannot write %s", path); return GPT_ERR_IO; }
    const size_t w = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    if (w != out.size()) { gpt_set_error("gpt_save_exr: short write to %s", path); return GPT_ERR_IO; }
    return GPT_OK;
}

int gpt_save_pfm(const char *path, int32_t width, int32_t height, const float *rgb)
{
    if (!path || !rgb || width <= 0 || height <= 0) { gpt_set_error("gpt_save_pfm: invalid argument"); return GPT_ERR_INVALID_ARG; }
    if (!imageio::write_pfm(path, width, height, rgb)) { gpt_set_error("gpt_save_pfm: cannot write %s", path); return GPT_ERR_IO; }
    return GPT_OK;
}

// The decoders on their own (what the scene loader does per "diffuse": "<file>" / "infinite": "<file>").  A null buffer
// asks for the size only; a buffer that is too small is refused.
int gpt_decode_image8(const char *path, int32_t *width, int32_t *height, int32_t *components, unsigned char *pixels, int64_t capacity)
{
    try {
        if (!path || !width || !height || !components) { gpt_set_error("gpt_decode_image8: invalid argument"); return GPT_ERR_INVALID_ARG; }
        std::vector<unsigned char> rgba;
        int w = 0, h = 0, comp = 0;
        if (!imageio::read_any8(path, w, h, comp, rgba)) {
            gpt_set_error("gpt_decode_image8: cannot read %s as PNG, JPEG, BMP, PNM or TGA", path);
            return GPT_ERR_IO;
        }
        *width = w; *height = h; *components = comp;
        if (!pixels) return GPT_OK;
        if (capacity < (int64_t)w * h * comp) { gpt_set_error("gpt_decode_image8: buffer too small"); return GPT_ERR_INVALID_ARG; }
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const unsigned char *p = &rgba[((size_t)(h - 1 - y) * w + x) * 4];       // stbi_set_flip_vertically_on_load(true)
                unsigned char *q = pixels + ((size_t)y * w + x) * comp;
                if (comp == 1) q[0] = p[0];
                else if (comp == 2) { q[0] = p[0]; q[1] = p[3]; }
                else for (int c = 0; c < comp; ++c) q[c] = p[c];
            }
        return GPT_OK;
    } catch (const std::exception &e) {
        gpt_set_error("gpt_decode_image8: %s", e.what());
        return GPT_ERR_IO;
    }
}

int gpt_load_texture(const char *path, int32_t *width, int32_t *height, gpt_uchar4 *texels, int64_t capacity)
{
    try {
        if (!path || !width || !height) { gpt_set_error("gpt_load_texture: invalid argument"); return GPT_ERR_INVALID_ARG; }
        std::vector<gpt_uchar4> t;
        int w = 0, h = 0;
        if (!imageio::load_texture(path, w, h, t)) { gpt_set_error("gpt_load_texture: cannot read %s as PNG, JPEG, BMP, PNM or TGA", path); return GPT_ERR_IO; }
        *width = w; *height = h;
        if (!texels) return GPT_OK;
        if (capacity < (int64_t)w * h) { gpt_set_error("gpt_load_texture: buffer too small"); return GPT_ERR_INVALID_ARG; }
        std::memcpy(texels, t.data(), t.size() * sizeof(gpt_uchar4));
        return GPT_OK;
    } catch (const std::exception &e) {
        gpt_set_error("gpt_load_texture: %s", e.what());
        return GPT_ERR_IO;
    }
}

int gpt_load_exr(const char *path, int32_t *width, int32_t *height, float *rgb, int64_t capacity)
{
    try {
        if (!path || !width || !height) { gpt_set_error("gpt_load_exr: invalid argument"); return GPT_ERR_INVALID_ARG; }
        std::vector<gpt_float3> px;
        int w = 0, h = 0;
        if (!imageio::read_exr_rgb_top_down(path, w, h, px)) { gpt_set_error("gpt_load_exr: cannot read %s (scanline OpenEXR: NONE, RLE, ZIPS, ZIP, PIZ)", path); return GPT_ERR_IO; }
        *width = w; *height = h;
        if (!rgb) return GPT_OK;
        if (capacity < (int64_t)w * h * 3) { gpt_set_error("gpt_load_exr: buffer too small"); return GPT_ERR_INVALID_ARG; }
        for (size_t i = 0; i < px.size(); ++i) { rgb[3 * i] = px[i].x; rgb[3 * i + 1] = px[i].y; rgb[3 * i + 2] = px[i].z; }
        return GPT_OK;
    } catch (const std::exception &e) {
        gpt_set_error("gpt_load_exr: %s", e.what());
        return GPT_ERR_IO;
    }
}

}  // extern "C"
