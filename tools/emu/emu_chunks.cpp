// emu_chunks.cpp -- TEST INFRASTRUCTURE: runs the kernels of csrc/chunks.hip on the CPU (tools/emu/hip/hip_runtime.h): the wave-parallel
// CRC-32 of crc32.hpp over a buffer, the chunk lexer (walk -> chunks -> finish) over one PNG file, and the IDAT writer.  From a prepared
// copy of the source (EMU_CHUNKS_SRC: launches blanked); built and used by tests/test_emu_chunks.py, never part of the product.
//
//   emu_chunks crc <file> <offset> <length> <running crc (hex)>        prints the CRC-32 of file[offset .. offset + length) (hex)
//   emu_chunks lex <png file> <idat out file> <list capacity> <waves>  prints status chunks aux0 aux1 width height depth color interlace ios idat_len consumed
//   emu_chunks write <stream file> <chunk bytes> <out file>            the IDAT chunks of the stream
#include EMU_CHUNKS_SRC

#include <fstream>
#include <iostream>
#include <vector>

using namespace spng;

static std::vector<uint8_t> slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const char *path, const uint8_t *p, size_t n) { std::ofstream f(path, std::ios::binary); f.write((const char *)p, (std::streamsize)n); }

static uint32_t g_out;
static void crc_probe(const uint8_t *p, uint64_t n, uint32_t crc)
{
    __shared__ uint32_t tab[CRC_TAB];
    const int lane = threadIdx.x;
    crc_table(tab, lane);
    const uint32_t c = wave_crc32(tab, (const gbyte *)p, n, crc, lane);
    if (lane == 0) g_out = c;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    if (mode == "crc" && argc >= 6) {
        std::vector<uint8_t> d = slurp(argv[2]);
        const uint64_t off = strtoull(argv[3], nullptr, 10), len = strtoull(argv[4], nullptr, 10);
        const uint32_t crc = (uint32_t)strtoul(argv[5], nullptr, 16);
        d.resize(d.size() + 64);
        emu::launch(1, 64, [&] { crc_probe(d.data() + off, len, crc); });
        printf("%08x\n", g_out);
        return 0;
    }
    if (mode == "lex" && argc >= 6) {
        std::vector<uint8_t> png = slurp(argv[2]);
        const size_t n = png.size();
        png.resize(n + 64);
        const uint32_t cap = (uint32_t)atoi(argv[4]), waves = (uint32_t)atoi(argv[5]);
        std::vector<uint8_t> idat(n + 64, 0xEE);
        spng_file_desc f{png.data(), n, idat.data(), getenv("EMU_IDAT_CAP") ? (uint64_t)atoll(getenv("EMU_IDAT_CAP")) : (uint64_t)n};
        spng_lexed out;
        memset(&out, 0, sizeof out);
        std::vector<LexChunk> table(cap + 1);
        uint64_t table_at[2] = {0, cap};
        LexWalk walk;
        emu::launch(1, 64, [&] { lex_walk_kernel(&f, &out, table.data(), table_at, &walk); });
        emu::launch(waves, 64, [&] { lex_chunk_kernel(&f, table.data(), table_at, &walk); }, 1);
        emu::launch(1, 64, [&] { lex_finish_kernel(&out, table.data(), table_at, &walk, &f, 1); });
        printf("%d %u %llx %llx %u %u %u %u %u %u %llu %llu\n", out.status, out.chunks, (unsigned long long)out.aux[0], (unsigned long long)out.aux[1], out.width,
               out.height, out.depth, out.color, out.interlace, out.ios, (unsigned long long)out.idat_len, (unsigned long long)out.consumed);
        spit(argv[3], idat.data(), (size_t)out.idat_len);
        return 0;
    }
    if (mode == "write" && argc >= 5) {
        std::vector<uint8_t> z = slurp(argv[2]);
        const uint64_t n = z.size(), piece = strtoull(argv[3], nullptr, 10);
        z.resize(n + 64);
        const uint64_t pieces = n ? (n + piece - 1) / piece : 0;
        std::vector<uint8_t> out(n + 12 * pieces + 64, 0xEE);
        spng_chunking_desc d{z.data(), n, out.data(), n + 12 * pieces, piece};
        spng_result res;
        emu::launch(3, 64, [&] { write_idat_kernel(&d, &res); }, 1);
        if (res.status != SPNG_DONE || res.written != n + 12 * pieces) { printf("bad result %d %llu\n", res.status, (unsigned long long)res.written); return 1; }
        spit(argv[4], out.data(), (size_t)res.written);
        printf("ok %llu\n", (unsigned long long)res.written);
        return 0;
    }
    return 2;
}
