// ThreadSanitizer harness for the multi-threaded frame builder (CPU only; the device half comes from tools/emu's stand-in runtime):
//   g++ -std=c++17 -O1 -g -fsanitize=thread -DSWFR_BUILD -I tools/emu/include -x c++ tools/tsan_frame_builder.cpp swf_renderer_amd/csrc/{renderer,frame_builder,geometry,shape_decoder}.cpp swf_renderer_amd/csrc/raster2.hip tools/emu/emu_rt.cpp -o build/tsan_harness -ldl -lpthread && SWFR_BUILD_THREADS=8 build/tsan_harness
#include <cstdio>
#include <cstring>
#include <vector>
#include "../include/swfr.h"
int main() {
    swfr_config cfg; std::memset(&cfg, 0, sizeof cfg); cfg.device = SWFR_DEVICE_HOST_ONLY;
    swfr_renderer* r = nullptr;
    if (swfr_create(640, 480, &cfg, &r) != 0) { std::puts("create failed"); return 1; }
    // one triangle shape
    swfr_fill_style fs; std::memset(&fs, 0, sizeof fs); fs.type = SWFR_FILL_SOLID; fs.color = swfr_rgba8{200, 10, 30, 128};
    swfr_shape_record rec[4]; std::memset(rec, 0, sizeof rec);
    rec[0].type = SWFR_RECORD_STYLE_CHANGE; rec[0].has_move_to = 1; rec[0].move_to_x = 100; rec[0].move_to_y = 100; rec[0].has_left_fill = 1; rec[0].left_fill = 1;
    rec[1].type = SWFR_RECORD_EDGE; rec[1].delta_x = 900; rec[1].delta_y = 200;
    rec[2].type = SWFR_RECORD_EDGE; rec[2].delta_x = -500; rec[2].delta_y = 700;
    rec[3].type = SWFR_RECORD_EDGE; rec[3].delta_x = -400; rec[3].delta_y = -900;
    swfr_define_shape tag; std::memset(&tag, 0, sizeof tag);
    tag.id = 1; tag.bounds = swfr_rect{0, 2000, 0, 2000};
    tag.initial_styles.n_fill = 1; tag.initial_styles.fill = &fs; tag.n_records = 4; tag.records = rec;
    uint32_t id = 0;
    if (swfr_register_shape(r, &tag, &id) != 0) { std::printf("register: %s\n", swfr_last_error(r)); return 1; }
    std::vector<swfr_display_object> kids;
    for (int frame = 0; frame < 300; ++frame) {
        // the child count -- and with it the number of pieces the pool builds -- changes from frame to frame
        static const size_t counts[5] = {1000, 130, 400, 70, 1000};
        kids.resize(counts[frame % 5]);
        for (size_t i = 0; i < kids.size(); ++i) {
            std::memset(&kids[i], 0, sizeof kids[i]);
            kids[i].type = SWFR_OBJECT_SHAPE; kids[i].id = id; kids[i].has_matrix = 1;
            kids[i].matrix.scale_x = 65536; kids[i].matrix.scale_y = 65536;
            kids[i].matrix.translate_x = int((i * 37 + frame * 11) % 11000) - (i < 100 ? 40000 : 0); kids[i].matrix.translate_y = int((i * 53) % 8000);
        }
        swfr_stage st; std::memset(&st, 0, sizeof st); st.n_children = uint32_t(kids.size()); st.children = kids.data();
        const swfr_edge* e; const swfr_path* p; const swfr_style* s; size_t ne, np, ns;
        if (swfr_build_frame(r, &st, &e, &ne, &p, &np, &s, &ns) != 0) { std::printf("build: %s\n", swfr_last_error(r)); return 1; }
        if (frame < 5) std::printf("edges %zu paths %zu styles %zu\n", ne, np, ns);
    }
    swfr_destroy(r);
    std::puts("tsan harness done");
    return 0;
}
