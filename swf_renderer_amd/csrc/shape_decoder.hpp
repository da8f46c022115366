// shape_decoder.hpp -- SWF shape records -> styled paths (host side, runs once per definition).
//
// Mirrors the reference's decoders:
//   decodeSwfShape       ts/src/lib/shape/decode-swf-shape.ts:22-39,298-448
//   decodeSwfMorphShape  ts/src/lib/shape/decode-swf-morph-shape.ts:21-41,265-425
// (fill0 gets the segment forward, fill1 reversed, the line forward; every `new_styles` opens a
// layer; per layer all fills in style order, then all lines; chains are built by one greedy pass.)
// Every coordinate is a [start,end] pair; static shapes use start == end.
#pragma once

#include <string>
#include <vector>

#include "../../include/swfr.h"

namespace swfr {

struct Coord {
    double s = 0, e = 0;  // start / end state (twips; halves appear for implied morph controls)
};

struct PathCommand {
    enum Kind { LineTo = 0, CurveTo = 1, MoveTo = 2 } kind;  // CommandType, ts/src/lib/shape/path.ts:4-8
    Coord cx, cy, x, y;
};

struct OwnedFill {
    swfr_fill_style style{};
    std::vector<swfr_color_stop> stops;
};

struct StyledPath {
    std::vector<PathCommand> commands;
    bool has_fill = false, has_line = false;
    OwnedFill fill;        // has_fill, or the line's fill when has_line
    uint32_t width = 0, morph_width = 0;
};

struct DecodedShape {
    bool morph = false;
    std::vector<StyledPath> paths;
};

// Throws std::runtime_error with the reference's messages ("Invalid fill ID", ...).
DecodedShape decode_shape(const swfr_define_shape& tag, bool morph);

// JSON.stringify(shape, null, 2) + "\n" as in the reference's decode goldens.
std::string shape_to_json(const DecodedShape& shape);

}  // namespace swfr
