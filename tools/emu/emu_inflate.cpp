// emu_inflate.cpp -- TEST INFRASTRUCTURE: runs inflate_kernel of csrc/inflate.hip (the serial three-wave kernel with the exact error
// vocabulary: scout / walker-decoder / resolver) on the CPU (tools/emu/hip/hip_runtime.h; host compiler clang++) over one stream and
// prints its result the way the oracle's is printed, so that tests/test_emu_inflate.py can compare status, counts, error payload and
// bytes.  From a prepared copy of the source (EMU_INFLATE_SRC); never part of the product.
//
//   emu_inflate <stream file> <format 0 zlib | 1 ios | 2 gzip-less raw> <capacity> <output file> [block_bit block_out token_bit token_out <bytes so far file>]
//   (the five: a call of spng_inflate_resume_batch that goes on where an earlier one stopped -- its four words of state and the output so far)
//   prints: status written consumed aux0 aux1
#include EMU_INFLATE_SRC

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
    if (argc < 5) { fprintf(stderr, "usage\n"); return 2; }
    std::vector<uint8_t> src = slurp(argv[1]);
    const int format = atoi(argv[2]);
    const uint64_t cap = strtoull(argv[3], nullptr, 10);
    const uint64_t n = src.size();
    src.resize(n + 64);
    std::vector<uint8_t> dst(cap + 64, 0xEE);
    uint64_t state[4] = {0, 0, 0, 0};
    InflateJob job;
    memset(&job, 0, sizeof job);
    job.src = src.data(); job.dst = dst.data(); job.src_len = n; job.dst_cap = cap; job.format = format; job.image = 0; job.skip = nullptr;
    job.state = nullptr; job.internal = 0;
    if (argc > 9) {
        for (int k = 0; k < 4; ++k) state[k] = strtoull(argv[5 + k], nullptr, 10);
        std::vector<uint8_t> sofar = slurp(argv[9]);
        memcpy(dst.data(), sofar.data(), sofar.size() < cap ? sofar.size() : cap);
        job.state = state;
    }
    spng_result res;
    memset(&res, 0xff, sizeof res);
    emu::launch(1, 256, [&] { inflate_kernel(&job, &res); });
    printf("%d %llu %llu %llu %llu\n", res.status, (unsigned long long)res.written, (unsigned long long)res.consumed,
           (unsigned long long)res.aux[0], (unsigned long long)res.aux[1]);
    std::ofstream o(argv[4], std::ios::binary);
    o.write((const char *)dst.data(), (std::streamsize)(res.written < cap ? res.written : cap));
    return 0;
}
