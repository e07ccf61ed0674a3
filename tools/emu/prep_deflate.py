"""TEST INFRASTRUCTURE: a copy of csrc/deflate.hip that the CPU wave emulator (tools/emu) can compile -- the product source is
not touched.  What is changed in the copy, each by an exact-text replacement that fails loudly when the source moves on:

  * kernel launches (`kernel<<<grid, block, 0, stream>>>(...)`) are blanked: the emulator's driver launches the kernels itself;
  * `s_waitcnt` inline assembly is dropped (memory is sequentially consistent in the emulator);
  * lock-step assumptions the fiber emulator does not keep (a wave's lanes run one after another between two wave builtins):
    a meeting of the wave is put where the source relies on "every lane stores, then every lane ORs", and a wave-uniform load
    that sits inside a lane-dependent `?:` is hoisted in front of it.

    python tools/emu/prep_deflate.py <csrc/deflate.hip> <out file>
"""
import re
import sys

MEET = '__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");'
PATCHES = [
    # bulk_put: the serial writer's pending byte is stored by every lane before any lane ORs its bits into the ring
    ("    if (b.nacc) s.out[b.total & (OUTB - 1)] = (uint8_t)b.acc;  // the pending bits of the serial writer join the ring\n",
     "    if (b.nacc) s.out[b.total & (OUTB - 1)] = (uint8_t)b.acc;\n    " + MEET + "\n"),
    # d2_relax_long: uni64 under a lane-dependent condition
    ("    const uint32_t w = (uint32_t)lane < cnt ? g.pool[uni64(g.bbase[q]) + pre + lane] : 0u;\n",
     "    const uint64_t emu_bb = uni64(g.bbase[q]);\n    const uint32_t w = (uint32_t)lane < cnt ? g.pool[emu_bb + pre + lane] : 0u;\n"),
]


def prepare(text: str, round_vertices: int = 0) -> str:
    """round_vertices: vertices per stream and round in the copy (a power of two; 0: the product's 2^21) -- small rounds let a
    20 KB input go through several rounds (state carried from round to round, the search a round ahead, both sets of records)"""
    if round_vertices:
        old = "static constexpr uint32_t D2_RV = 1u << 21;"
        assert old in text
        text = text.replace(old, f"static constexpr uint32_t D2_RV = {round_vertices}u;")
        old3 = "static constexpr uint32_t D3_RV = 1u << 21;"
        assert old3 in text
        text = text.replace(old3, f"static constexpr uint32_t D3_RV = {round_vertices}u;")
    out = text.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory")', "((void)0)")
    # "LDS serves a wave's operations in order": where the source only tells the COMPILER to keep an order (the keys of a group
    # of vertices before the next group's reads of the ring), the emulator's lanes have to meet
    assert out.count('asm volatile("" ::: "memory");') >= 2
    out = out.replace('asm volatile("" ::: "memory");', MEET)
    out, n = re.subn(r"\b[A-Za-z_][A-Za-z_0-9]*(?:<[A-Za-z_0-9, ]*>)?<<<[^;]*?>>>\([^;]*?\);", "(void)0;", out, flags=re.S)
    assert n >= 8, f"only {n} kernel launches found"
    for old, new in PATCHES:
        assert old in out, "deflate.hip moved on: " + old.strip()[:80]
        out = out.replace(old, new)
    return out


def prepare_plain(text: str, patches=()) -> str:
    """the same for a source without lock-step patches of its own (csrc/unfilter.hip): launches blanked, `s_waitcnt` dropped,
    compiler-only barriers turned into meetings of the wave, plus the exact-text `patches` given"""
    out = text.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory")', "((void)0)")
    out = out.replace('asm volatile("" ::: "memory");', MEET)
    out, n = re.subn(r"\b[A-Za-z_][A-Za-z_0-9]*(?:<[A-Za-z_0-9, ]*>)?<<<[^;]*?>>>\([^;]*?\);", "(void)0;", out, flags=re.S)
    assert n >= 1
    for old, new in patches:
        assert old in out, "the source moved on: " + old.strip()[:80]
        out = out.replace(old, new)
    return out


def prepare_inflate(text: str, huffman_text: str, huffman_path: str, common_path: str):
    """csrc/inflate.hip (the serial three-wave kernel) and the copy of huffman.hpp it is to include: `COMPILER_ORDER()` -- a barrier for
    the compiler only, where the hardware keeps a wave's LDS operations in order -- becomes a meeting of the wave; the eight-token
    chain walk written in GCN assembly (v_readlane pairs + s_bitset1) is restated in C; `s_endpgm` aborts.  -> (source, header)"""
    hh = huffman_text.replace('#define COMPILER_ORDER() asm volatile("" ::: "memory")', "#define COMPILER_ORDER() " + MEET.rstrip(";"))
    assert hh != huffman_text
    hh = hh.replace('#include "common.hpp"', f'#include "{common_path}"')
    a = text.index('                            asm volatile(\n                                "v_readlane_b32 %3, %10, %11')
    tail = ': "v"(nxt), "v"(nxt2), "s"(ent));'
    b = text.index(tail, a) + len(tail)
    walk = """                            {
                                auto RL = [&](uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(l & 63)); };
                                t2 = RL(nxt2, ent); t1 = RL(nxt, ent); t4 = RL(nxt2, t2); t3 = RL(nxt, t2);
                                t6 = RL(nxt2, t4); t5 = RL(nxt, t4); p = RL(nxt2, t6); t7 = RL(nxt, t6);
                                chain = 1ull << (ent & 63) | 1ull << (t1 & 63) | 1ull << (t2 & 63) | 1ull << (t3 & 63) | 1ull << (t4 & 63) |
                                        1ull << (t5 & 63) | 1ull << (t6 & 63) | 1ull << (t7 & 63);
                            }"""
    patches = [('asm volatile("s_endpgm")', "abort()"), (text[a:b], walk),
               ('asm("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(p));', "chain |= 1ull << (p & 63);"),
               ('#include "huffman.hpp"', f'#include "{huffman_path}"')]
    return prepare_plain(text, patches=patches), hh


if __name__ == "__main__":
    open(sys.argv[2], "w").write(prepare(open(sys.argv[1]).read()))
