// Sanitizer driver for everything in this tree that parses files or command lines: the hand-written ONNX protobuf reader
// (onnx_pb.cpp), constant folding and lowering (fold.cpp, lower.cpp), the engine-file reader (plan.cpp), the tile grid (tiles.cpp),
// the built-in image / video codecs (cli/imageio.cpp) and the option parser (cli/args.cpp).  `make asan` builds it with
// g++ -fsanitize=address,undefined -fno-sanitize-recover=all - host code only, no HIP - and tests/test_malformed_inputs.py feeds it
// truncated and bit-flipped files.  The reference has no such target (CMakeLists.txt:49-68).
//   w2x_parse_check onnx  FILE BATCH TILE [fp32]    load_onnx -> fold_graph -> lower_graph -> serialize -> deserialize
//   w2x_parse_check plan  FILE                      Plan::deserialize (+ validate), what load() runs on an engine file
//   w2x_parse_check image FILE [deep]               read_image, then write_image to /dev/null-like temp names in every format it allows
//   w2x_parse_check avi   FILE                      AviReader::open + every frame
//   w2x_parse_check args  ARGS...                   cli::parse
//   w2x_parse_check tiles W H T S TOUT OVX OVY      calculate_tiles
// Exit code: 0 = accepted, 2 = rejected with a message on stderr (the clean `false` of the product path); anything else is a crash
// or a sanitizer report.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../lower.h"
#include "../tiles.h"
#include "args.h"
#include "imageio.h"

using namespace w2x;

static int run(int argc, char** argv) {
    if (argc < 3) throw std::runtime_error("usage: w2x_parse_check {onnx|plan|image|avi|args|tiles} ...");
    const std::string mode = argv[1];
    if (mode == "onnx") {
        if (argc < 5) throw std::runtime_error("onnx FILE BATCH TILE [fp32]");
        const int batch = atoi(argv[3]), tile = atoi(argv[4]);
        Plan plan = build_plan(argv[2], batch, 3, tile, tile, argc > 5);
        plan.userB = batch;
        const std::vector<uint8_t> bytes = plan.serialize();
        const Plan back = Plan::deserialize(bytes.data(), bytes.size());
        printf("ok: %zu ops, %zu tensors, %zu bytes of engine file, %s\n", back.ops.size(), back.tensors.size(), bytes.size(), onnx_op_histogram(argv[2]).substr(0, 60).c_str());
    } else if (mode == "plan") {
        std::ifstream f(argv[2], std::ios::binary | std::ios::ate);
        if (!f.is_open()) throw std::runtime_error("could not open engine file");
        std::vector<char> bytes((size_t)f.tellg());
        f.seekg(0); f.read(bytes.data(), (std::streamsize)bytes.size());
        const Plan p = Plan::deserialize((const uint8_t*)bytes.data(), bytes.size());
        printf("ok: %zu ops\n", p.ops.size());
    } else if (mode == "image") {
        const cli::Bitmap b = cli::read_image(argv[2], argc > 3);
        const std::string tmp = std::string(argv[2]) + ".rewritten";
        cli::write_image(tmp + ".png", b);
        if (b.bgr16.empty()) { cli::write_image(tmp + ".bmp", b); cli::Bitmap c = b; c.alpha.clear(); cli::write_image(tmp + ".ppm", c); }
        printf("ok: %d x %d%s%s\n", b.cols, b.rows, b.alpha.empty() ? "" : " +alpha", b.bgr16.empty() ? "" : " 16-bit");
    } else if (mode == "avi") {
        cli::AviReader rd; std::string why;
        if (!rd.open(argv[2], &why)) throw std::runtime_error("not an AVI this reader takes: " + why);
        std::vector<uint8_t> frame((size_t)rd.info().width * rd.info().height * 3);
        int n = 0;
        while (rd.read(frame.data())) ++n;
        printf("ok: %d x %d, %d of %d frames\n", rd.info().width, rd.info().height, n, rd.info().frames);
    } else if (mode == "args") {
        const cli::Options o = cli::parse(argc - 1, argv + 1);
        printf("ok: %s\n", cli::to_json(o).c_str());
    } else if (mode == "tiles") {
        if (argc < 9) throw std::runtime_error("tiles W H T S TOUT OVX OVY");
        const int W = atoi(argv[2]), H = atoi(argv[3]), T = atoi(argv[4]), S = atoi(argv[5]), TO = atoi(argv[6]);
        const TileGrid g = calculate_tiles(W, H, W * S, H * S, T, T, TO, TO, S, atof(argv[7]), atof(argv[8]));
        for (int parts = 1; parts <= 8 && parts <= g.nx; ++parts) for (int p = 0; p < parts; ++p) (void)strip_plan(g, W * S, TO, p, parts);
        printf("ok: %d tiles (%d x %d)\n", g.count, g.nx, g.ny);
    } else throw std::runtime_error("unknown mode " + mode);
    return 0;
}

int main(int argc, char** argv) {
    try { return run(argc, argv); }
    catch (const std::exception& e) { fprintf(stderr, "rejected: %s\n", e.what()); return 2; }
}
