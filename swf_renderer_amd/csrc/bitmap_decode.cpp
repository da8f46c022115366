// bitmap_decode.cpp -- `image/x-swf-bmp` (DefineBitsLossless, format 3: zlib-compressed colour-mapped image) -> straight RGBA8.
// Host-side restatement of decodeXSwfBmpSync (ts/src/lib/decode-x-swf-bmp.ts:9-41), the decoder behind
// NodeCanvasBitmapService.addBitmap (ts/src/lib/renderers/node-canvas-bitmap-service.ts:14-37): one format byte (3), width and
// height (u16 LE), colour count - 1 (u8), then a zlib stream holding the RGB palette followed by the indices, rows padded to 4
// bytes; every palette colour is opaque, an index past the palette is opaque black.  The inflater below is a plain RFC 1950 / 1951
// decoder (stored, fixed and dynamic Huffman blocks; Adler-32 checked), so that libswfr.so needs no zlib at link time.
#include "bitmap_decode.hpp"

#include <cstring>

namespace swfr {
namespace {

struct BitReader {
    const uint8_t* p; size_t n, pos = 0; uint32_t bitbuf = 0; int bitcnt = 0;
    bool need(int k) {
        while (bitcnt < k) {
            if (pos >= n) return false;
            bitbuf |= uint32_t(p[pos++]) << bitcnt; bitcnt += 8;
        }
        return true;
    }
    bool bits(int k, uint32_t& v) {
        if (k == 0) { v = 0; return true; }
        if (!need(k)) return false;
        v = bitbuf & ((1u << k) - 1u); bitbuf >>= k; bitcnt -= k;
        return true;
    }
};

// canonical Huffman code: count[len] codes of each length, symbols in code order
struct Huffman {
    uint16_t count[16]; uint16_t symbol[288];
    bool build(const uint8_t* lengths, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) ++count[lengths[i]];
        if (count[0] == n) return true;                        // no codes: legal for an unused distance table
        int left = 1;
        for (int len = 1; len < 16; ++len) { left <<= 1; left -= count[len]; if (left < 0) return false; }   // over-subscribed
        uint16_t offs[16]; offs[1] = 0;
        for (int len = 1; len < 15; ++len) offs[len + 1] = uint16_t(offs[len] + count[len]);
        for (int i = 0; i < n; ++i) if (lengths[i]) symbol[offs[lengths[i]]++] = uint16_t(i);
        return true;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; ++len) {
            uint32_t b;
            if (!br.bits(1, b)) return -1;
            code |= int(b);
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

bool inflate_codes(BitReader& br, std::vector<uint8_t>& out, size_t cap, const Huffman& lit, const Huffman& dist) {
    for (;;) {
        const int sym = lit.decode(br);
        if (sym < 0) return false;
        if (sym < 256) { if (out.size() >= cap) return false; out.push_back(uint8_t(sym)); continue; }
        if (sym == 256) return true;
        const int li = sym - 257;
        if (li >= 29) return false;
        uint32_t eb;
        if (!br.bits(LEN_EXTRA[li], eb)) return false;
        const size_t len = LEN_BASE[li] + eb;
        const int ds = dist.decode(br);
        if (ds < 0 || ds >= 30) return false;
        if (!br.bits(DIST_EXTRA[ds], eb)) return false;
        const size_t d = DIST_BASE[ds] + eb;
        if (d > out.size() || out.size() + len > cap) return false;
        for (size_t i = 0; i < len; ++i) out.push_back(out[out.size() - d]);
    }
}

}  // namespace

bool zlib_inflate(const uint8_t* data, size_t len, size_t max_out, std::vector<uint8_t>& out) {
    out.clear();
    if (len < 6) return false;
    // RFC 1950 header: deflate, window <= 32 KiB, check bits, no preset dictionary
    if ((data[0] & 0x0f) != 8 || (data[0] >> 4) > 7 || ((uint32_t(data[0]) << 8) | data[1]) % 31 != 0 || (data[1] & 0x20)) return false;
    BitReader br{data + 2, len - 2};
    for (;;) {
        uint32_t last, type;
        if (!br.bits(1, last) || !br.bits(2, type)) return false;
        if (type == 0) {
            br.bitbuf = 0; br.bitcnt = 0;                      // to the next byte boundary
            if (br.pos + 4 > br.n) return false;
            const uint32_t n = br.p[br.pos] | (uint32_t(br.p[br.pos + 1]) << 8), nn = br.p[br.pos + 2] | (uint32_t(br.p[br.pos + 3]) << 8);
            br.pos += 4;
            if ((n ^ nn) != 0xffffu || br.pos + n > br.n || out.size() + n > max_out) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + n);
            br.pos += n;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 144; ++i) lengths[i] = 8;
                for (int i = 144; i < 256; ++i) lengths[i] = 9;
                for (int i = 256; i < 280; ++i) lengths[i] = 7;
                for (int i = 280; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint32_t hlit, hdist, hclen;
                if (!br.bits(5, hlit) || !br.bits(5, hdist) || !br.bits(4, hclen)) return false;
                const int nlen = int(hlit) + 257, ndist = int(hdist) + 1, ncode = int(hclen) + 4;
                if (nlen > 286 || ndist > 30) return false;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) { uint32_t v; if (!br.bits(3, v)) return false; cl[ORDER[i]] = uint8_t(v); }
                Huffman lencode;
                if (!lencode.build(cl, 19)) return false;
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = lencode.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) { lengths[idx++] = uint8_t(sym); continue; }
                    uint8_t prev = 0; uint32_t rep;
                    if (sym == 16) { if (idx == 0) return false; prev = lengths[idx - 1]; if (!br.bits(2, rep)) return false; rep += 3; }
                    else if (sym == 17) { if (!br.bits(3, rep)) return false; rep += 3; }
                    else { if (!br.bits(7, rep)) return false; rep += 11; }
                    if (idx + int(rep) > nlen + ndist) return false;
                    while (rep--) lengths[idx++] = prev;
                }
                if (lengths[256] == 0) return false;               // no end-of-block code
                if (!lit.build(lengths, nlen) || !dist.build(lengths + nlen, ndist)) return false;
            }
            if (!inflate_codes(br, out, max_out, lit, dist)) return false;
        } else return false;
        if (last) break;
    }
    // Adler-32 of the output, big endian, behind the last (possibly partial) byte of the deflate stream
    if (br.pos + 4 > br.n) return false;
    const uint8_t* t = br.p + br.pos;
    const uint32_t want = (uint32_t(t[0]) << 24) | (uint32_t(t[1]) << 16) | (uint32_t(t[2]) << 8) | t[3];
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < out.size(); ++i) { a = (a + out[i]) % 65521u; b = (b + a) % 65521u; }
    return ((b << 16) | a) == want;
}

XSwfBmpStatus decode_x_swf_bmp(const uint8_t* data, size_t len, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba) {
    if (len < 6) return XSwfBmpStatus::Corrupt;
    if (data[0] != 3) return XSwfBmpStatus::UnsupportedFormat;           // decode-x-swf-bmp.ts:12-14 UnsupportedXSwfBmpFormatId
    width = data[1] | (uint32_t(data[2]) << 8);
    height = data[3] | (uint32_t(data[4]) << 8);
    const size_t padded = width + ((4 - (width % 4)) % 4);               // :17
    const size_t n_colors = size_t(data[5]) + 1, table = 3 * n_colors;   // :18, :25
    std::vector<uint8_t> src;
    if (!zlib_inflate(data + 6, len - 6, table + padded * height + 65536, src)) return XSwfBmpStatus::Corrupt;
    if (src.size() < table + (height ? padded * (height - 1) + width : 0)) return XSwfBmpStatus::Corrupt;
    rgba.assign(size_t(width) * height * 4, 0);
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            const size_t ci = src[table + y * padded + x];
            uint8_t* px = &rgba[(size_t(y) * width + x) * 4];
            if (ci < n_colors) { px[0] = src[3 * ci]; px[1] = src[3 * ci + 1]; px[2] = src[3 * ci + 2]; }    // :26-31
            px[3] = 255;                                                 // (:35: an index past the palette is opaque black)
        }
    return XSwfBmpStatus::Ok;
}

}  // namespace swfr
