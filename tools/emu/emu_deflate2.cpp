// emu_deflate2.cpp -- TEST INFRASTRUCTURE: runs the round kernels of csrc/deflate.hip (dfl2_begin / dfl3_search /
// dfl2_advance / dfl2_parse) on the CPU (tools/emu/hip/hip_runtime.h) over one stream, round by round and with the two sets
// of candidate records alternating the way api.hip drives them, and compares the bytes with the expected stream (the oracle's).
// Built and used by tests/test_emu_deflate.py from a copy of deflate.hip whose launchers and `s_waitcnt` lines are blanked
// (EMU_DEFLATE_SRC); never part of the product.
//
//   emu_deflate2 <input file> <expected stream file> <level> <format 0 zlib | 1 raw (ios)> [chunks per stream] [cut ...]
//   cut ...: the input arrives in pieces (spng_deflate_resume_batch): a call per cut with `more` set, the state kept, then the rest
//   exit code 0: SPNG_DONE and identical bytes; 1: anything else (printed)
#include EMU_DEFLATE_SRC

#include <fstream>
#include <iostream>

using namespace spng;

static std::vector<uint8_t> slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage\n"); return 2; }
    std::vector<uint8_t> src = slurp(argv[1]), want = slurp(argv[2]);
    const int level = atoi(argv[3]), format = atoi(argv[4]);
    const uint32_t cps = argc > 5 ? (uint32_t)atoi(argv[5]) : 3;
    const uint64_t n = src.size();
    std::vector<uint8_t> dst(want.size() + 4096, 0xEE);
    src.resize(n + 64);                                          // (the kernels read keys a few bytes past positions they search)

    if (level < 8 && getenv("EMU_ONE_KERNEL")) {
        // levels 0-7, the one-kernel form (SPNG_DEFLATE_ONE_KERNEL): one wave per stream does everything
        std::vector<uint32_t> ring(65536, 0);
        DeflateJob j;
        memset(&j, 0, sizeof j);
        j.src = src.data(); j.dst = dst.data(); j.src_len = n; j.dst_cap = dst.size(); j.ring = ring.data();
        j.format = format == 1 ? SPNG_FORMAT_IOS : SPNG_FORMAT_ZLIB; j.level = level; j.image = 0; j.exponent = 15;
        spng_result r1;
        memset(&r1, 0xff, sizeof r1);
        emu::launch(1, 64, [&] { deflate_kernel(&j, &r1); });
        if (r1.status != SPNG_DONE) { printf("status %d\n", r1.status); return 1; }
        if (r1.written != want.size() || memcmp(dst.data(), want.data(), want.size())) {
            size_t k = 0;
            while (k < want.size() && k < r1.written && dst[k] == want[k]) ++k;
            printf("stream differs: %llu bytes against %zu expected, first difference at %zu\n", (unsigned long long)r1.written, want.size(), k);
            return 1;
        }
        printf("ok: %llu -> %llu bytes (greedy / lazy kernel)\n", (unsigned long long)n, (unsigned long long)r1.written);
        return 0;
    }
    if (level < 8) {
        // levels 0-7 in rounds: dfl3_begin -> (dfl3_search_fast + dfl3_advance, dfl3_parse) per round, the two sets of answers
        // alternating; `cut ...`: a call per piece with `more` set and the state kept, as spng_deflate_resume_batch drives it
        const uint64_t RV = deflate3_round_positions();
        std::vector<uint32_t> match[2] = {std::vector<uint32_t>(RV + 64, 0xDEADBEEF), std::vector<uint32_t>(RV + 64, 0xBEEFDEAD)};
        std::vector<uint8_t> statebuf(deflate_state_bytes(), 0);
        D3Stream st;
        memset(&st, 0, sizeof st);
        st.src = src.data(); st.dst = dst.data(); st.src_len = n; st.dst_cap = dst.size();
        st.format = format == 1 ? SPNG_FORMAT_IOS : SPNG_FORMAT_ZLIB; st.level = level; st.image = 0; st.exponent = 15;
        st.state = (D1State *)statebuf.data();
        st.match[0] = match[0].data(); st.match[1] = match[1].data();
        spng_result res;
        memset(&res, 0xff, sizeof res);
        const uint32_t chunk = (uint32_t)(((RV / cps + 63) / 64) * 64);
        std::vector<uint64_t> cuts;
        for (int a = 6; a < argc; ++a) cuts.push_back(strtoull(argv[a], nullptr, 10));
        cuts.push_back(n);
        // one-shot streams: the blocks side by side (dfl4_walk / dfl4_block / dfl4_scan / dfl4_place), as api.hip drives them;
        // EMU_TWO_WAVE forces the two-wave parse (the form of streams that arrive in pieces)
        const bool blocks = cuts.size() == 1 && !getenv("EMU_TWO_WAVE");
        const uint32_t maxb = (uint32_t)deflate4_max_blocks(RV);
        std::vector<uint32_t> terms(RV + 64 + 4096, 0xABABABAB), bdesc(4 + 2 * (size_t)maxb, 0);
        std::vector<uint64_t> bbits(2 * (size_t)maxb, 0);
        std::vector<uint8_t> scratch(blocks ? (size_t)maxb * deflate4_block_bytes() + 64 : 64, 0xCD);
        if (blocks) { st.terms = terms.data(); st.bdesc = bdesc.data(); st.bbits = bbits.data(); st.scratch = scratch.data(); }
        uint32_t rounds = 0;
        uint64_t spos = 0;
        for (size_t call = 0; call < cuts.size(); ++call) {
            st.src_len = cuts[call] < n ? cuts[call] : n;
            st.more = call + 1 < cuts.size() ? 1 : 0;
            const uint64_t E = deflate3_end(st.src_len, st.more != 0), span = E > spos ? E - spos : 0;
            const uint32_t calls_rounds = span ? (uint32_t)((span + RV - 1) / RV) : 1;
            emu::launch(1, 256, [&] { dfl3_begin_kernel(&st, 1); });
            for (uint32_t r = 0; r < calls_rounds; ++r, ++rounds) {
                const uint32_t par = r & 1;
                emu::launch(cps, SPNG_D3_WAVES * 64, [&] { dfl3_search_fast_kernel(&st, cps, chunk, par); });
                emu::launch(1, 256, [&] { dfl3_advance_kernel(&st, 1); });
                if (blocks) {
                    emu::launch(1, 64, [&] { dfl4_walk_kernel(&st, par); });
                    if (getenv("EMU_DUMP_TERMS")) {
                        std::ofstream o(getenv("EMU_DUMP_TERMS"), std::ios::binary);
                        const uint32_t nbk = bdesc[0]; uint32_t tot = 0;
                        for (uint32_t k = 0; k < nbk; ++k) tot += bdesc[5 + 2 * k] & 0x7fffffffu;
                        o.write((const char *)&nbk, 4); o.write((const char *)bdesc.data(), 4 * (4 + 2 * (size_t)nbk));
                        o.write((const char *)&tot, 4); o.write((const char *)terms.data(), 4 * (size_t)tot);
                        const uint32_t nm = (uint32_t)(n + 1); o.write((const char *)&nm, 4); o.write((const char *)match[par].data(), 4 * (size_t)nm);
                    }
                    // (the product launches the worst case of blocks -- the host does not know their number -- and the idle ones return at
                    //  once; here only the ones the walk made, + 1 idle: a workgroup costs the emulator 64 fibers)
                    const uint32_t grid = bdesc[0] + 1 < maxb ? bdesc[0] + 1 : maxb;
                    emu::launch(grid, 64, [&] { dfl4_block_kernel(&st); });
                    emu::launch(1, 256, [&] { dfl4_scan_kernel(&st, maxb, &res); });
                    emu::launch(grid, 64, [&] { dfl4_place_kernel(&st, maxb); });
                } else emu::launch(1, 128, [&] { dfl3_parse_kernel(&st, &res, par); });
            }
            if (st.more && res.status != SPNG_NEED_MORE_INPUT) { printf("call %zu (more): status %d\n", call, res.status); return 1; }
            if (st.more) { if (res.aux[1] != st.state->spos) { printf("call %zu: aux[1] %llu != spos %llu\n", call, (unsigned long long)res.aux[1], (unsigned long long)st.state->spos); return 1; } spos = res.aux[1]; }
        }
        if (res.status != SPNG_DONE || !st.state->done) { printf("status %d done %u after %u rounds\n", res.status, st.state->done, rounds); return 1; }
        if (res.written != want.size() || memcmp(dst.data(), want.data(), want.size())) {
            size_t k = 0;
            while (k < want.size() && k < res.written && dst[k] == want[k]) ++k;
            printf("stream differs: %llu bytes against %zu expected, first difference at %zu\n", (unsigned long long)res.written, want.size(), k);
            return 1;
        }
        printf("ok: %llu -> %llu bytes in %u rounds (search + parse kernels%s)\n", (unsigned long long)n, (unsigned long long)res.written, rounds, blocks ? ", blocks side by side" : "");
        return 0;
    }
    const uint64_t V = deflate2_vertices(n), B = V / 64 + 2;
    std::vector<uint16_t> vinfo[2] = {std::vector<uint16_t>(V, 0xDEAD), std::vector<uint16_t>(V, 0xBEEF)};
    std::vector<uint64_t> bbase[2] = {std::vector<uint64_t>(B, 0), std::vector<uint64_t>(B, 0)};
    std::vector<uint32_t> bwords[2] = {std::vector<uint32_t>(B, 0), std::vector<uint32_t>(B, 0)};
    std::vector<uint64_t> emask(B, 0);
    std::vector<uint32_t> up(V + 2, 0), step(V + 2, 0);
    std::vector<uint8_t> pathb(V + 2, 0), litb(B, 0);
    const uint64_t pool_words = (n < (1u << 21) ? n : (1u << 21)) * 30 + 4096;
    std::vector<uint32_t> pool[2] = {std::vector<uint32_t>(pool_words, 0), std::vector<uint32_t>(pool_words, 0)};
    unsigned long long pool_next[2] = {0, 0};
    std::vector<uint32_t> rings((size_t)cps * SPNG_D3_WAVES * 30 * 64, 0);    // (the search workgroups' word scratch)
    D2State state;
    memset(&state, 0, sizeof state);
    D2Stream st;
    memset(&st, 0, sizeof st);
    st.src = src.data(); st.dst = dst.data(); st.src_len = n; st.dst_cap = dst.size();
    st.format = format == 1 ? SPNG_FORMAT_IOS : SPNG_FORMAT_ZLIB; st.level = level; st.image = 0; st.exponent = 15; st.more = 0;
    st.state = &state;
    st.vinfo = vinfo[0].data(); st.bbase = bbase[0].data(); st.bwords = bwords[0].data(); st.emask = emask.data();
    st.vinfo2 = vinfo[1].data(); st.bbase2 = bbase[1].data(); st.bwords2 = bwords[1].data();
    st.up = up.data(); st.step = step.data(); st.pathb = pathb.data(); st.litb = litb.data();
    spng_result res;
    memset(&res, 0xff, sizeof res);

    uint64_t pos = 0; uint32_t lim = 2048;
    const uint32_t chunk = ((D2_RV / cps + 63) / 64) * 64;
    std::vector<uint64_t> cuts;
    for (int a = 6; a < argc; ++a) cuts.push_back(strtoull(argv[a], nullptr, 10));
    cuts.push_back(n);
    uint32_t rounds = 0;
    for (size_t call = 0; call < cuts.size(); ++call) {
        // one call of spng_deflate_resume_batch: the bytes so far, `more` unless they are all
        st.src_len = cuts[call] < n ? cuts[call] : n;
        st.more = call + 1 < cuts.size() ? 1 : 0;
        const uint32_t calls_rounds = deflate2_plan(st.src_len, st.more != 0, pos, lim);
        emu::launch(1, 256, [&] { dfl2_begin_kernel(&st, 1); });
        for (uint32_t r = 0; r < calls_rounds; ++r, ++rounds) {
            const uint32_t par = r & 1;
            pool_next[par] = 0;
            emu::launch(cps, SPNG_D3_WAVES * 64, [&] { dfl3_search_kernel(&st, cps, chunk, pool[par].data(), &pool_next[par], pool_words, rings.data(), par); });
            emu::launch(1, 256, [&] { dfl2_advance_kernel(&st, 1); });
            emu::launch(1, 64, [&] { dfl2_parse_kernel(&st, pool[par].data(), &res, par); });
            if (getenv("EMU_VERBOSE")) fprintf(stderr, "call %zu round %u: pos %llu limit %u total %llu fail %u done %u status %d\n", call, r, (unsigned long long)state.pos,
                                               state.limit, (unsigned long long)state.total, state.fail, state.done, res.status);
        }
        if (st.more && res.status != SPNG_NEED_MORE_INPUT) { printf("call %zu (more): status %d\n", call, res.status); return 1; }
    }
    if (getenv("EMU_DUMP")) { std::ofstream o(getenv("EMU_DUMP"), std::ios::binary); o.write((const char *)dst.data(), (std::streamsize)res.written); }
    if (res.status != SPNG_DONE || !state.done || state.fail) { printf("status %d done %u fail %u after %u rounds\n", res.status, state.done, state.fail, rounds); return 1; }
    if (res.written != want.size() || memcmp(dst.data(), want.data(), want.size())) {
        size_t k = 0;
        while (k < want.size() && k < res.written && dst[k] == want[k]) ++k;
        printf("stream differs: %llu bytes against %zu expected, first difference at %zu\n", (unsigned long long)res.written, want.size(), k);
        return 1;
    }
    printf("ok: %llu -> %llu bytes in %u rounds\n", (unsigned long long)n, (unsigned long long)res.written, rounds);
    return 0;
}
