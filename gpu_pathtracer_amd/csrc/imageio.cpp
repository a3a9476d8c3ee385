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
// output), a PNG reader (8-bit grey / grey+alpha / RGB / RGBA / palette,
// non-interlaced) on a small inflate, and PFM (little- or big-endian float32)
// for linear radiance and environment maps.  JPEG and OpenEXR decoding are not
// implemented yet (DESIGN.md "Next").
#include "imageio.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
bool inflate_raw(const unsigned char *src, size_t n, std::vector<unsigned char> &out)
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
                if (sym < 256) out.push_back((unsigned char)sym);
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return false;
                    int len = lbase[sym] + br.bits(lext[sym]);
                    int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    size_t d = (size_t)dbase[ds] + (size_t)br.bits(dext[ds]);
                    if (d > out.size()) return false;
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
    FILE *f = std::fopen(path, "rb");
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

// 8-bit PNG -> RGBA8, rows top-down.  components = channels in the file (1, 2, 3 or 4; palette -> 3/4)
bool read_png(const char *path, int &width, int &height, int &components, std::vector<unsigned char> &rgba)
{
    std::vector<unsigned char> d;
    if (!read_file(path, d) || d.size() < 33 || std::memcmp(d.data(), "\x89PNG\r\n\x1a\n", 8) != 0) return false;
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte, trns;
    width = height = 0;
    while (pos + 12 <= d.size()) {
        uint32_t len = be32(&d[pos]);
        const unsigned char *type = &d[pos + 4];
        if (pos + 12 + len > d.size()) return false;
        const unsigned char *body = &d[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            width = (int)be32(body); height = (int)be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    if (width <= 0 || height <= 0 || depth != 8 || interlace != 0 || idat.size() < 6) return false;
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    std::vector<unsigned char> raw;
    if (!inflate_raw(idat.data() + 2, idat.size() - 2, raw)) return false;
    const size_t stride = (size_t)width * ch;
    if (raw.size() < (stride + 1) * (size_t)height) return false;
    std::vector<unsigned char> img(stride * (size_t)height);
    for (int y = 0; y < height; ++y) {
        const unsigned char *in = &raw[(stride + 1) * (size_t)y];
        unsigned char *cur = &img[stride * (size_t)y];
        const unsigned char *up = y ? &img[stride * (size_t)(y - 1)] : nullptr;
        int ft = in[0];
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= (size_t)ch ? cur[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int x = in[i + 1];
            int v = ft == 0 ? x : ft == 1 ? x + a : ft == 2 ? x + b : ft == 3 ? x + ((a + b) >> 1) : ft == 4 ? x + paeth(a, b, c) : -1;
            if (v < 0) return false;
            cur[i] = (unsigned char)v;
        }
    }
    components = ctype == 3 ? (trns.empty() ? 3 : 4) : ch;
    rgba.resize((size_t)width * height * 4);
    for (size_t i = 0; i < (size_t)width * height; ++i) {
        unsigned char r, g, b, a = 255;
        if (ctype == 0) { r = g = b = img[i]; }
        else if (ctype == 4) { r = g = b = img[2 * i]; a = img[2 * i + 1]; }
        else if (ctype == 2) { r = img[3 * i]; g = img[3 * i + 1]; b = img[3 * i + 2]; }
        else if (ctype == 6) { r = img[4 * i]; g = img[4 * i + 1]; b = img[4 * i + 2]; a = img[4 * i + 3]; }
        else {
            size_t k = img[i];
            if (3 * k + 2 >= plte.size()) return false;
            r = plte[3 * k]; g = plte[3 * k + 1]; b = plte[3 * k + 2];
            if (k < trns.size()) a = trns[k];
        }
        rgba[4 * i] = r; rgba[4 * i + 1] = g; rgba[4 * i + 2] = b; rgba[4 * i + 3] = a;
    }
    return true;
}

// ImageIO::LoadTexture + Texture::Texture (src/imageio.cpp:11-59, src/texture.h:15-27):
// flip vertically, 1/255, sRGB -> linear by powf(x, 2.2f) on r,g,b, then truncate x*255 back to uchar.
bool load_texture_png(const char *path, int &width, int &height, std::vector<gpt_uchar4> &texels)
{
    std::vector<unsigned char> rgba;
    int comp = 0;
    if (!read_png(path, width, height, comp, rgba)) return false;
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

int gpt_save_pfm(const char *path, int32_t width, int32_t height, const float *rgb)
{
    if (!path || !rgb || width <= 0 || height <= 0) { gpt_set_error("gpt_save_pfm: invalid argument"); return GPT_ERR_INVALID_ARG; }
    if (!imageio::write_pfm(path, width, height, rgb)) { gpt_set_error("gpt_save_pfm: cannot write %s", path); return GPT_ERR_IO; }
    return GPT_OK;
}

}  // extern "C"
