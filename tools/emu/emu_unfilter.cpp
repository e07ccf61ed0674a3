// emu_unfilter.cpp -- TEST INFRASTRUCTURE: runs the scanline kernels of csrc/unfilter.hip (unfilter_pk_kernel<4 | 8>, the line-aligned
// kernel of the 4- and 8-byte pixel formats, and the byte-wise unfilter_kernel<1 | 2 | 3 | 6>) on the CPU (tools/emu/hip/hip_runtime.h; host compiler: clang++, for the kernel's vector
// extensions) over one image and compares the rows with the expected ones (the oracle's).  Built and used by
// tests/test_emu_unfilter.py from a prepared copy of the source (EMU_UNFILTER_SRC); never part of the product.
//
//   emu_unfilter <filtered scanlines file> <expected rows file> <pitch bytes> <rows> <bpp 1|2|3|4|6|8> <rows per piece>
//   exit code 0: identical; 1: not (printed)
#include EMU_UNFILTER_SRC

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
    std::vector<uint8_t> in = slurp(argv[1]), want = slurp(argv[2]);
    const uint32_t pitch = (uint32_t)atoi(argv[3]), rows = (uint32_t)atoi(argv[4]), bpp = (uint32_t)atoi(argv[5]), piece_rows = (uint32_t)atoi(argv[6]);
    if (in.size() != (size_t)rows * (pitch + 1) || want.size() != (size_t)rows * pitch) { fprintf(stderr, "sizes\n"); return 2; }
    in.resize(in.size() + 64);
    std::vector<uint8_t> out(want.size() + 64, 0xEE);
    UnfJob job;
    memset(&job, 0, sizeof job);
    job.in = in.data(); job.out = out.data(); job.in_stride = pitch + 1; job.out_stride = pitch; job.stream_off = 0; job.rows_len = nullptr;
    job.pitch = pitch; job.rows = rows; job.image = 0; job.bpp = bpp; job.has_prev = 0;
    const uint32_t np = (rows + piece_rows - 1) / piece_rows;
    if (bpp == 4) emu::launch(1, SPNG_UNF_PK_NW * 64, [&] { unfilter_pk_kernel<4>(&job, nullptr, piece_rows, np); }, (np + Pk<4>::NCH - 1) / Pk<4>::NCH);
    else if (bpp == 8) emu::launch(1, SPNG_UNF_PK_NW * 64, [&] { unfilter_pk_kernel<8>(&job, nullptr, piece_rows, np); }, (np + Pk<8>::NCH - 1) / Pk<8>::NCH);
    // the byte-wise kernel of the other pixel sizes (sub-byte depths defilter with bpp 1): a workgroup per piece
    else if (bpp == 1 && job.pitch >= 2048) emu::launch(1, SPNG_UNF_NW * 64, [&] { unfilter_kernel<4, 1, 32>(&job, nullptr, piece_rows); }, np);
    else if (bpp == 1) emu::launch(1, SPNG_UNF_NW * 64, [&] { unfilter_kernel<4, 1>(&job, nullptr, piece_rows); }, np);
    else if (bpp == 2 && job.pitch >= 2048) emu::launch(1, SPNG_UNF_NW * 64, [&] { unfilter_kernel<4, 2, 32>(&job, nullptr, piece_rows); }, np);
    else if (bpp == 2) emu::launch(1, SPNG_UNF_NW * 64, [&] { unfilter_kernel<4, 2>(&job, nullptr, piece_rows); }, np);
    else if (bpp == 3) emu::launch(1, SPNG_UNF_NW * 64, [&] { unfilter_kernel<3>(&job, nullptr, piece_rows); }, np);
    else if (bpp == 6) emu::launch(1, SPNG_UNF_NW * 64, [&] { unfilter_kernel<6>(&job, nullptr, piece_rows); }, np);
    else { fprintf(stderr, "bpp\n"); return 2; }
    if (memcmp(out.data(), want.data(), want.size())) {
        size_t k = 0;
        while (k < want.size() && out[k] == want[k]) ++k;
        printf("rows differ: first difference at byte %zu = row %zu column %zu (filter %u)\n", k, k / pitch, k % pitch, in[(k / pitch) * (pitch + 1)]);
        return 1;
    }
    printf("ok: %u rows of %u bytes in %u pieces\n", rows, pitch, np);
    return 0;
}
