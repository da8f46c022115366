// bitmap_decode.hpp -- `image/x-swf-bmp` decoding on the host (see bitmap_decode.cpp).
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace swfr {

// RFC 1950 zlib stream -> bytes (at most max_out of them); false: corrupt, truncated, or larger than max_out
bool zlib_inflate(const uint8_t* data, size_t len, size_t max_out, std::vector<uint8_t>& out);

enum class XSwfBmpStatus { Ok, UnsupportedFormat, Corrupt };
// decodeXSwfBmpSync (ts/src/lib/decode-x-swf-bmp.ts:9-41): format 3 -> width, height, straight RGBA8 with tight rows
XSwfBmpStatus decode_x_swf_bmp(const uint8_t* data, size_t len, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba);

}  // namespace swfr
