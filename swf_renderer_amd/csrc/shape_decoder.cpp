// shape_decoder.cpp -- see shape_decoder.hpp.
#include "shape_decoder.hpp"

#include <charconv>
#include <cmath>
#include <deque>
#include <stdexcept>

namespace swfr {
namespace {

struct Segment {
    bool curved = false;
    Coord sx, sy, cx, cy, ex, ey;
};

struct StyleBucket {
    OwnedFill fill;
    uint32_t width = 0, morph_width = 0;
    std::vector<Segment> segments;
};

struct Layer {
    std::vector<StyleBucket> fills, lines;
};

OwnedFill own_fill(const swfr_fill_style& s, bool morph) {
    OwnedFill f;
    f.style = s;
    if (morph && s.type != SWFR_FILL_SOLID) throw std::runtime_error("Unknown fill type");  // decode-swf-morph-shape.ts:94-106
    if (s.type > SWFR_FILL_BITMAP) throw std::runtime_error("UnknownFillStyle");
    if (s.n_stops && s.stops) f.stops.assign(s.stops, s.stops + s.n_stops);
    f.style.stops = nullptr;
    f.style.n_stops = uint32_t(f.stops.size());
    if (!morph) f.style.morph_color = f.style.color;
    return f;
}

Layer make_layer(const swfr_styles& st, bool morph) {
    Layer l;
    for (uint32_t i = 0; i < st.n_fill; ++i) {
        StyleBucket b;
        b.fill = own_fill(st.fill[i], morph);
        l.fills.push_back(std::move(b));
    }
    for (uint32_t i = 0; i < st.n_line; ++i) {
        StyleBucket b;
        b.fill = own_fill(st.line[i].fill, morph);
        b.width = st.line[i].width;
        b.morph_width = morph ? st.line[i].morph_width : st.line[i].width;
        l.lines.push_back(std::move(b));
    }
    return l;
}

// extractContinuous: one greedy pass over the remaining segments, matching on START-state coordinates.
std::deque<Segment> take_chain(std::vector<Segment>& open) {
    std::deque<Segment> chain;
    chain.push_back(open.front());
    open.erase(open.begin());
    double sx = chain.front().sx.s, sy = chain.front().sy.s, ex = chain.front().ex.s, ey = chain.front().ey.s;
    for (size_t i = 0; i < open.size();) {
        const Segment& c = open[i];
        if (c.sx.s == ex && c.sy.s == ey) {
            ex = c.ex.s;
            ey = c.ey.s;
            chain.push_back(c);
            open.erase(open.begin() + i);
        } else if (c.ex.s == sx && c.ey.s == sy) {
            sx = c.sx.s;
            sy = c.sy.s;
            chain.push_front(c);
            open.erase(open.begin() + i);
        } else {
            ++i;
        }
    }
    return chain;
}

std::vector<PathCommand> to_commands(const std::vector<Segment>& segments) {
    std::vector<Segment> open = segments;
    std::vector<PathCommand> out;
    while (!open.empty()) {
        std::deque<Segment> chain = take_chain(open);
        PathCommand mv{};
        mv.kind = PathCommand::MoveTo;
        mv.x = chain.front().sx;
        mv.y = chain.front().sy;
        out.push_back(mv);
        for (const Segment& s : chain) {
            PathCommand c{};
            c.kind = s.curved ? PathCommand::CurveTo : PathCommand::LineTo;
            c.cx = s.cx;
            c.cy = s.cy;
            c.x = s.ex;
            c.y = s.ey;
            out.push_back(c);
        }
    }
    return out;
}

void layer_paths(const Layer& l, std::vector<StyledPath>& out) {
    for (const StyleBucket& b : l.fills) {
        std::vector<PathCommand> cmds = to_commands(b.segments);
        if (cmds.empty()) continue;
        StyledPath p;
        p.commands = std::move(cmds);
        p.has_fill = true;
        p.fill = b.fill;
        out.push_back(std::move(p));
    }
    for (const StyleBucket& b : l.lines) {
        std::vector<PathCommand> cmds = to_commands(b.segments);
        if (cmds.empty()) continue;
        StyledPath p;
        p.commands = std::move(cmds);
        p.has_line = true;
        p.fill = b.fill;
        p.width = b.width;
        p.morph_width = b.morph_width;
        out.push_back(std::move(p));
    }
}

}  // namespace

DecodedShape decode_shape(const swfr_define_shape& tag, bool morph) {
    std::vector<Layer> layers;
    int left = -1, right = -1, line = -1;  // indices into the current layer, -1 = none
    Coord x, y;

    auto open_layer = [&](const swfr_styles& st) {
        layers.push_back(make_layer(st, morph));
        left = right = line = -1;
    };
    auto pick = [&](uint32_t id, size_t count) -> int {
        if (id == 0) return -1;
        if (id - 1 >= count) throw std::runtime_error("Invalid fill ID");
        return int(id - 1);
    };

    open_layer(tag.initial_styles);
    for (uint32_t i = 0; i < tag.n_records; ++i) {
        const swfr_shape_record& r = tag.records[i];
        if (r.type == SWFR_RECORD_STYLE_CHANGE) {
            // order matters: newStyles, leftFill, rightFill, lineStyle, moveTo (decode-swf-shape.ts:337-356);
            // morph shapes never carry new styles (decode-swf-morph-shape.ts:304-322)
            if (!morph && r.new_styles) open_layer(*r.new_styles);
            if (r.has_left_fill) left = pick(r.left_fill, layers.back().fills.size());
            if (r.has_right_fill) right = pick(r.right_fill, layers.back().fills.size());
            if (r.has_line_style) line = pick(r.line_style, layers.back().lines.size());
            if (r.has_move_to) {
                if (morph && !r.has_morph_move_to) throw std::runtime_error("Expected morphMoveTo to be defined");
                x = Coord{double(r.move_to_x), morph ? double(r.morph_move_to_x) : double(r.move_to_x)};
                y = Coord{double(r.move_to_y), morph ? double(r.morph_move_to_y) : double(r.move_to_y)};
            }
        } else if (r.type == SWFR_RECORD_EDGE) {
            const double mdx = morph ? r.morph_delta_x : r.delta_x, mdy = morph ? r.morph_delta_y : r.delta_y;
            const Coord ex{x.s + r.delta_x, x.e + mdx}, ey{y.s + r.delta_y, y.e + mdy};
            Segment fwd, rev;
            const bool has_c = r.has_control_delta, has_mc = morph ? r.has_morph_control_delta : r.has_control_delta;
            if (has_c || has_mc) {
                // a missing control delta is delta/2, un-floored (decode-swf-morph-shape.ts:341-346)
                const double cdx = has_c ? r.control_delta_x : r.delta_x / 2.0, cdy = has_c ? r.control_delta_y : r.delta_y / 2.0;
                double mcx, mcy;
                if (!morph) {
                    mcx = cdx;
                    mcy = cdy;
                } else {
                    mcx = r.has_morph_control_delta ? r.morph_control_delta_x : mdx / 2.0;
                    mcy = r.has_morph_control_delta ? r.morph_control_delta_y : mdy / 2.0;
                }
                fwd.curved = rev.curved = true;
                fwd.cx = rev.cx = Coord{x.s + cdx, x.e + mcx};
                fwd.cy = rev.cy = Coord{y.s + cdy, y.e + mcy};
            }
            fwd.sx = x; fwd.sy = y; fwd.ex = ex; fwd.ey = ey;
            rev.sx = ex; rev.sy = ey; rev.ex = x; rev.ey = y;
            Layer& cur = layers.back();
            if (left >= 0) cur.fills[left].segments.push_back(fwd);
            if (right >= 0) cur.fills[right].segments.push_back(rev);
            if (line >= 0) cur.lines[line].segments.push_back(fwd);
            x = ex;
            y = ey;
        } else {
            throw std::runtime_error("UnreachableCode");
        }
    }
    DecodedShape out;
    out.morph = morph;
    for (const Layer& l : layers) layer_paths(l, out.paths);
    return out;
}

// ---------------------------------------------------------------------------------------------
// JSON in the shape.ts.json layout
// ---------------------------------------------------------------------------------------------
namespace {
std::string js_number(double v) {
    if (v == std::floor(v) && std::fabs(v) < 1e15) return std::to_string((long long)v);
    char buf[64];
    auto res = std::to_chars(buf, buf + sizeof buf, v);  // shortest round-trip, as JS prints
    return std::string(buf, res.ptr);
}

class Json {
public:
    void open(char c) { put_value_prefix(); out_ += c; stack_.push_back(0); }
    void close(char c) {
        const int n = stack_.back();
        stack_.pop_back();
        if (n) { out_ += '\n'; indent(); }
        out_ += c;
    }
    void key(const char* k) {
        sep();
        out_ += '"'; out_ += k; out_ += "\": ";
        keyed_ = true;
    }
    void num(double v) { put_value_prefix(); out_ += js_number(v); }
    void boolean(bool b) { put_value_prefix(); out_ += b ? "true" : "false"; }
    void pair(const Coord& c) { open('['); num(c.s); num(c.e); close(']'); }
    std::string take() { return std::move(out_); }

private:
    void indent() { out_.append(2 * stack_.size(), ' '); }
    void sep() {
        if (stack_.back()++) out_ += ',';
        out_ += '\n';
        indent();
    }
    void put_value_prefix() {
        if (keyed_) { keyed_ = false; return; }
        if (!stack_.empty()) sep();
    }
    std::string out_;
    std::vector<int> stack_;
    bool keyed_ = false;
};

void json_color(Json& j, const swfr_rgba8& c) {
    j.open('{');
    j.key("r"); j.num(c.r / 255.0);
    j.key("g"); j.num(c.g / 255.0);
    j.key("b"); j.num(c.b / 255.0);
    j.key("a"); j.num(c.a / 255.0);
    j.close('}');
}
void json_matrix(Json& j, const swfr_matrix& m) {
    auto eps = [&](const char* k, int32_t v) { j.key(k); j.open('{'); j.key("epsilons"); j.num(v); j.close('}'); };
    j.open('{');
    eps("scaleX", m.scale_x); eps("scaleY", m.scale_y); eps("rotateSkew0", m.rotate_skew0); eps("rotateSkew1", m.rotate_skew1);
    j.key("translateX"); j.num(m.translate_x);
    j.key("translateY"); j.num(m.translate_y);
    j.close('}');
}
// FillStyleType (ts/src/lib/shape/fill-style.ts:5-10): Bitmap 0, FocalGradient 1, LinearGradient 2, Solid 3
void json_fill(Json& j, const OwnedFill& f, bool morph) {
    const swfr_fill_style& s = f.style;
    j.open('{');
    if (morph) {
        j.key("type"); j.num(0);  // MorphFillStyleType.Solid
        j.key("startColor"); json_color(j, s.color);
        j.key("endColor"); json_color(j, s.morph_color);
    } else if (s.type == SWFR_FILL_SOLID) {
        j.key("type"); j.num(3);
        j.key("color"); json_color(j, s.color);
    } else if (s.type == SWFR_FILL_BITMAP) {
        j.key("type"); j.num(0);
        j.key("bitmapId"); j.num(s.bitmap_id);
        j.key("matrix"); json_matrix(j, s.matrix);
        j.key("repeating"); j.boolean(s.repeating);
        j.key("smoothed"); j.boolean(s.smoothed);
    } else {
        j.key("type"); j.num(s.type == SWFR_FILL_LINEAR_GRADIENT ? 2 : 1);
        j.key("matrix"); json_matrix(j, s.matrix);
        j.key("gradient");
        j.open('{');
        j.key("colors");
        j.open('[');
        for (const swfr_color_stop& st : f.stops) {
            j.open('{');
            j.key("ratio"); j.num(st.ratio / 255.0);
            j.key("color"); json_color(j, st.color);
            j.close('}');
        }
        j.close(']');
        j.close('}');
        if (s.type != SWFR_FILL_LINEAR_GRADIENT) {
            j.key("focalPoint");
            j.num(s.type == SWFR_FILL_FOCAL_GRADIENT ? s.focal_point / 256.0 : 0.0);
        }
    }
    j.close('}');
}
}  // namespace

std::string shape_to_json(const DecodedShape& shape) {
    Json j;
    const bool m = shape.morph;
    auto coord = [&](const char* k, const Coord& c) {
        j.key(k);
        if (m) j.pair(c); else j.num(c.s);
    };
    j.open('{');
    j.key("paths");
    j.open('[');
    for (const StyledPath& p : shape.paths) {
        j.open('{');
        j.key("commands");
        j.open('[');
        for (const PathCommand& c : p.commands) {
            j.open('{');
            j.key("type"); j.num(int(c.kind));
            if (c.kind == PathCommand::MoveTo) {
                coord("x", c.x); coord("y", c.y);
            } else {
                if (c.kind == PathCommand::CurveTo) { coord("controlX", c.cx); coord("controlY", c.cy); }
                coord("endX", c.x); coord("endY", c.y);
            }
            j.close('}');
        }
        j.close(']');
        if (p.has_fill) { j.key("fill"); json_fill(j, p.fill, m); }
        if (p.has_line) {
            j.key("line");
            j.open('{');
            j.key("width");
            if (m) j.pair(Coord{double(p.width), double(p.morph_width)}); else j.num(p.width);
            j.key("fill"); json_fill(j, p.fill, m);
            j.close('}');
        }
        j.close('}');
    }
    j.close(']');
    j.close('}');
    return j.take() + "\n";
}

}  // namespace swfr
