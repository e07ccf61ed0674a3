// emu_pinflate2.cpp -- TEST INFRASTRUCTURE: runs the kernels of csrc/pinflate2.hip on the CPU (tools/emu/hip/hip_runtime.h)
// over one DEFLATE stream, the way api.hip drives them (find -> decode -> scan -> resolve), and compares the bytes with the
// expected output.  Built and used by tests/test_emu_pinflate.py; never part of the product.
//
//   emu_pinflate2 <stream file> <expected output file> <format 0|1> <segment bytes> [pool pages] [resume: start_bit out_pos]
//   exit code 0: the pipeline produced SPNG_DONE and identical bytes;  3: the pipeline left the whole stream to the serial
//   kernel;  4: it decoded a prefix (printed: the block boundary and byte count the serial kernel would start from) and that
//   prefix is right;  5: it reported an error of its own (printed);  1: wrong bytes / wrong result
#include "../../swift_png_amd/csrc/pinflate2.hip"

#include <fstream>
#include <iostream>

using namespace spng;

static std::vector<uint8_t> slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

#ifdef SPNG_EMU_COV
static void cov_print() { fprintf(stderr, "COV"); for (int i = 0; i < 6; ++i) fprintf(stderr, " %ld", spng::g_cov[i]); fprintf(stderr, "\n"); }
#endif
int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage\n"); return 2; }
#ifdef SPNG_EMU_COV
    atexit(cov_print);
#endif
    std::vector<uint8_t> src = slurp(argv[1]), want = slurp(argv[2]);
    const int format = atoi(argv[3]);
    uint64_t seg_bytes = strtoull(argv[4], nullptr, 10);
    seg_bytes = (seg_bytes + 255) & ~(uint64_t)255;
    const uint32_t pages = argc > 5 ? (uint32_t)atoi(argv[5]) : 4096;
    const bool verbose = getenv("EMU_VERBOSE") != nullptr;
    const uint32_t ptcap = getenv("EMU_PTCAP") ? (uint32_t)atoi(getenv("EMU_PTCAP")) : pages;   // page-table entries per segment

    std::vector<uint8_t> dst(want.size() + 64, 0xEE);
    src.resize(src.size());
    PStream st;
    memset(&st, 0, sizeof st);
    st.src = src.data(); st.dst = dst.data(); st.src_len = src.size(); st.dst_cap = want.size();
    st.format = format; st.image = 0;
    uint64_t state[4] = {0, 0, 0, 0};                          // (four words: the pipeline moves the first pair and clears the second)
    st.state = state;                                          // (api.hip: every stream has a state slot, {0, 0} unless resumed)
    if (argc > 7) {
        state[0] = st.start_bit = strtoull(argv[6], nullptr, 10); state[1] = st.out_pos = strtoull(argv[7], nullptr, 10);
        memcpy(dst.data(), want.data(), st.out_pos);
    }
    const bool resumed = argc > 7;
    uint64_t k = (src.size() + seg_bytes - 1) / seg_bytes;
    if (k < 1) k = 1;
    st.seg_first = 0; st.seg_count = (uint32_t)k; st.seg_bytes = seg_bytes;
    std::vector<PSeg> segs(k);
    uint64_t pt_total = 0;
    for (uint64_t q = 0; q < k; ++q) {
        PSeg &sg = segs[q];
        memset(&sg, 0, sizeof sg);
        sg.stream = 0; sg.index = (uint32_t)q;
        sg.log_cap = ptcap;
        sg.log_off = pt_total; pt_total += sg.log_cap;
        sg.start_bit = ~0ull;
    }
    std::vector<uint32_t> pt(pt_total, 0xDEADBEEF);
    std::vector<uint8_t> poolmem((size_t)pages << PAGE_SHIFT, 0xAB);
    uint32_t nextv[4] = {0, 0, 0, 0};                          // (pages taken, blocks decoded)
    uint32_t &next = nextv[0];
    DPool pool{poolmem.data(), nextv, pages, 0};
    spng_result res;
    memset(&res, 0xff, sizeof res);
    int32_t done = 0;

    emu::launch((unsigned)k, 64, [&] { pinf2_find_kernel<0>(&st, segs.data(), 0); });
    if (verbose) for (uint64_t q = 0; q < k; ++q) fprintf(stderr, "seg %llu: start %lld\n", (unsigned long long)q, (long long)segs[q].start_bit);
    emu::launch((unsigned)k, 64, [&] { pinf2_decode_kernel<0>(&st, segs.data(), pt.data(), pool, 0); });
    if (verbose) for (uint64_t q = 0; q < k; ++q)
        fprintf(stderr, "seg %llu: end %lld status %d nhw %llu next %u\n", (unsigned long long)q, (long long)segs[q].end_bit, segs[q].status,
                (unsigned long long)segs[q].ntok, segs[q].next);
    // EMU_PARTS=n: the stream's chain in up to n parts, resolved by n workgroups (what api.hip does for batches of few streams)
    const uint32_t pmax = getenv("EMU_PARTS") ? (uint32_t)atoi(getenv("EMU_PARTS")) : 0;
    std::vector<PPart> parts(pmax ? pmax : 1);
    memset(parts.data(), 0, parts.size() * sizeof(PPart));
    std::vector<uint16_t> sym(pmax ? want.size() + 4096 + 64 : 8, 0xEEEE);
    std::vector<uint8_t> win((size_t)(pmax ? pmax : 1) * 32768, 0xCD);
    st.parts_max = pmax; st.sym_off = 0;
    emu::launch(1, 64, [&] { pinf2_scan_kernel<0>(&st, segs.data(), parts.data()); });
    if (verbose && pmax) for (uint32_t q = 0; q < st.parts; ++q)
        fprintf(stderr, "part %u: seg %u .. %u out %llu + %llu\n", q, parts[q].seg, parts[q].seg_end, (unsigned long long)parts[q].out_pos, (unsigned long long)parts[q].out_len);
    if (verbose) fprintf(stderr, "stream: ok %d nhw %llu end_bit %llu pages used %u\n", st.ok, (unsigned long long)st.ntok, (unsigned long long)st.end_bit, next);
    {   // the token stream, expanded the plain way: tells a decode bug from a resolve bug
        std::vector<uint8_t> out(st.out_pos ? std::vector<uint8_t>(want.begin(), want.begin() + st.out_pos) : std::vector<uint8_t>());
        uint32_t sk = 0; bool okt = st.ok != 0;
        for (uint32_t hops = 0; okt && hops < k; ++hops) {
            const PSeg &sg = segs[sk];
            for (uint64_t i = 0; i < sg.ntok; ++i) {
                auto hw = [&](uint64_t idx) -> uint32_t {
                    const uint64_t u = idx >> 3; const uint32_t pid = pt[sg.log_off + (u >> (PAGE_SHIFT - 4))];
                    return ((const uint16_t *)(poolmem.data() + ((size_t)pid << PAGE_SHIFT) + ((u & (PAGE_UNITS - 1)) << 4)))[idx & 7];
                };
                const uint32_t v = hw(i);
                if (!(v & 0x8000)) out.push_back((uint8_t)v);
                else if ((v & 0xC000) == 0x8000) {
                    const uint32_t h1 = hw(i + 1), run = (v & 0xff) + 3, d = (((v >> 8) & 63) | (h1 & 0x1ff) << 6) + 1;
                    if ((h1 & 0xC000) != 0xC000 || d > out.size()) { fprintf(stderr, "token stream: bad reference at hw %llu of seg %u\n", (unsigned long long)i, sk); okt = false; break; }
                    for (uint32_t r = 0; r < run; ++r) out.push_back(out[out.size() - d]);
                    ++i;
                }
            }
            if (sg.status == PSEG_FINAL || sg.status == PSEG_PARTIAL) break;
            sk = sg.next;
        }
        size_t i = 0;
        while (i < out.size() && i < want.size() && out[i] == want[i]) ++i;
        if (verbose || i != out.size() || (st.ok == 1 && out.size() != want.size()))
            fprintf(stderr, "token stream expands to %zu bytes, agrees with the expected bytes up to %zu of %zu\n", out.size(), i, want.size());
    }
    emu::launch(1, RT2, [&] { pinf2_resolve_kernel<0, false, true>(&st, segs.data(), pt.data(), pool, &res, &done, parts.data(), pmax, nullptr); });
    // EMU_RETRY_PAGES=n: a stream whose segments found the pool empty takes the retry pass (api.hip: the pool to itself and
    // its like -- here a second pool of n pages)
    if (getenv("EMU_RETRY_PAGES")) {
        printf("first pass: ok %d pass %u\n", st.ok, st.pass);
        const uint32_t pages2 = (uint32_t)atoi(getenv("EMU_RETRY_PAGES"));
        std::vector<uint8_t> poolmem2((size_t)pages2 << PAGE_SHIFT, 0xAB);
        uint32_t next2v[4] = {0, 0, 0, 0};
        uint32_t &next2 = next2v[0];
        DPool pool2{poolmem2.data(), next2v, pages2, 0};
        emu::launch((unsigned)k, 64, [&] { pinf2_find_kernel<1>(&st, segs.data(), 0); });
        emu::launch((unsigned)k, 64, [&] { pinf2_decode_kernel<1>(&st, segs.data(), pt.data(), pool2, 0); });
        emu::launch(1, 64, [&] { pinf2_scan_kernel<1>(&st, segs.data(), parts.data()); });
        emu::launch(1, RT2, [&] { pinf2_resolve_kernel<1, false, true>(&st, segs.data(), pt.data(), pool2, &res, &done, parts.data(), 0, nullptr); });
        printf("retry pass: ok %d done %d pages %u\n", st.ok, done, next2);
    }
    if (pmax >= 2) {
        // (api.hip picks the marker parts' geometry by their number; here EMU_MARK_TILE=8192 asks for the big tiles)
        if (getenv("EMU_MARK_TILE") && atoi(getenv("EMU_MARK_TILE")) == 8192)
            emu::launch(pmax - 1, RT2, [&] { pinf2_resolve_kernel<0, true, true>(&st, segs.data(), pt.data(), pool, &res, &done, parts.data(), pmax, sym.data()); });
        else
            emu::launch(pmax - 1, RT2, [&] { pinf2_resolve_kernel<0, true, false>(&st, segs.data(), pt.data(), pool, &res, &done, parts.data(), pmax, sym.data()); });
        emu::launch(1, 512, [&] { pinf2_window_kernel(&st, parts.data(), pmax, sym.data(), win.data()); });
        emu::launch(FIX_WG, 256, [&] { pinf2_fixup_kernel(&st, parts.data(), pmax, sym.data(), win.data()); }, pmax - 1);
        emu::launch(1, 64, [&] { pinf2_verdict_kernel(&st, parts.data(), pmax, &res, &done, 1); });
        printf("parts: %u\n", st.parts);
    }

    if (resumed) {
        // report what the pipeline did
        printf("resume ok=%d done=%d state=%llu,%llu\n", st.ok, done, (unsigned long long)state[0], (unsigned long long)state[1]);
        const uint64_t upto = st.ok == 2 ? state[1] : (done ? res.written : 0);
        if (memcmp(dst.data(), want.data(), upto) != 0) { printf("MISMATCH in the resumed prefix\n"); return 1; }
        return 0;
    }
    if (!done && (state[0] || state[1])) {
        printf("partial: the serial kernel starts at bit %llu with %llu bytes in front of it\n", (unsigned long long)state[0], (unsigned long long)state[1]);
        if (state[1] > want.size() || memcmp(dst.data(), want.data(), state[1]) != 0) { printf("MISMATCH in the prefix\n"); return 1; }
        return 4;
    }
    if (done && res.status != SPNG_DONE) {
        printf("error %d aux %llx %llx written %llu consumed %llu\n", res.status, (unsigned long long)res.aux[0], (unsigned long long)res.aux[1],
               (unsigned long long)res.written, (unsigned long long)res.consumed);
        return 5;
    }
    if (!done) {
        size_t i = 0;
        while (i < want.size() && dst[i] == want[i]) ++i;
        printf("declined (ok=%d); output agrees with the expected bytes up to %zu of %zu\n", st.ok, i, want.size());
        return 3;
    }
    if (res.status != SPNG_DONE || res.reserved != 1 || res.written != want.size()) {
        printf("bad result: status %d written %llu (want %zu)\n", res.status, (unsigned long long)res.written, want.size());
        return 1;
    }
    for (size_t i = 0; i < want.size(); ++i)
        if (dst[i] != want[i]) { printf("MISMATCH at byte %zu: %02x != %02x\n", i, dst[i], want[i]); return 1; }
    for (size_t i = want.size(); i < dst.size(); ++i)
        if (dst[i] != 0xEE) { printf("wrote past the end at %zu\n", i); return 1; }
    printf("ok: %zu -> %zu bytes, %llu segments, %u pages, consumed %llu\n", src.size(), want.size(), (unsigned long long)k, next,
           (unsigned long long)res.consumed);
    return 0;
}
