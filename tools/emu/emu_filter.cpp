// emu_filter.cpp -- TEST INFRASTRUCTURE: runs filter_kernel of csrc/encode.hip (PNG.Encoder.filter: the five residuals of a scanline
// scored, the first strict minimum kept) on the CPU (tools/emu/hip/hip_runtime.h; host compiler clang++) over one non-interlaced
// image and compares the filtered scanlines with the expected ones (the oracle's).  From a prepared copy of the source
// (EMU_FILTER_SRC); never part of the product.
//
//   emu_filter <storage file> <expected scanlines file> <width> <height> <depth> <channels>
#include EMU_FILTER_SRC

#include <fstream>
#include <iostream>
#include <vector>

using namespace spng;

static std::vector<uint8_t> slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage\n"); return 2; }
    std::vector<uint8_t> storage = slurp(argv[1]), want = slurp(argv[2]);
    const uint32_t w = (uint32_t)atoi(argv[3]), h = (uint32_t)atoi(argv[4]), depth = (uint32_t)atoi(argv[5]), channels = (uint32_t)atoi(argv[6]);
    const uint32_t pitch = (w * depth * channels + 7) / 8;
    if (want.size() != (size_t)h * (pitch + 1)) { fprintf(stderr, "sizes\n"); return 2; }
    storage.resize(storage.size() + 64);
    std::vector<uint8_t> rows(want.size() + 64, 0xEE);
    FilterJob job;
    memset(&job, 0, sizeof job);
    job.storage = storage.data(); job.rows = rows.data(); job.row_stride = pitch + 1; job.sub_w = w; job.sub_h = h; job.width = w;
    job.bx = 0; job.by = 0; job.sx = 1; job.sy = 1; job.depth = depth; job.channels = channels; job.pitch = pitch;
    const unsigned bx = (h + 3) / 4 ? (h + 3) / 4 : 1;
    emu::launch(bx > 8 ? 8 : bx, 256, [&] { filter_kernel(&job); }, 1);
    if (memcmp(rows.data(), want.data(), want.size())) {
        size_t k = 0;
        while (k < want.size() && rows[k] == want[k]) ++k;
        printf("scanlines differ: first difference at byte %zu = row %zu column %zu\n", k, k / (pitch + 1), k % (pitch + 1));
        return 1;
    }
    printf("ok: %u rows of %u bytes\n", h, pitch);
    return 0;
}
