// fg_fuse.hpp -- FRAMING INSIDE THE DECODE KERNELS (round 6): what a wave does with a byte TILE of the raw stream so that the lines
// it decodes come out of its own delimiter scan instead of an offsets array written by an earlier kernel.
//
// Replaces, together with the decoders' stage B, the whole body of LineSplitter::run / NulSplitter::run for a chunk of the stream
//   `for line in buf_reader.lines()` + the "Invalid UTF-8 input" rejection      src/flowgger/splitter/line_splitter.rs:17-25
//   `for line in buf_reader.split(0)` + `str::from_utf8`                          src/flowgger/splitter/nul_splitter.rs:18-40
// with ONE read of the stream (fg_frame.hip framed in a pass of its own: the stream was read twice and the frame count visited the host
// between the two kernels).
//
// Geometry.  The stream is cut into TILES of S bytes (S a multiple of 16, the launcher's choice: about as many bytes as the format's
// group of lines holds).  A wave stages  [T * S - 16, T * S + S + look)  -- the tile, the 16 bytes in front of it (is the byte before
// the tile a terminator?) and `look` bytes behind it (where the tile's last line ends) -- and OWNS the lines that START inside the
// tile.  Tile positions x are relative to the staged range (x = 16 <=> stream byte T * S).  Per 16-byte chunk c the stage-A code keeps
// one word  m32[c] = delimiter mask | UTF-8 error mask << 16  in LDS (chunk_masks below, the classifiers of fg_frame.hip).
//   * count_tile:  every lane owns a run of consecutive chunks: delimiters that start one of the tile's lines (x in [15, own_end)) and
//     all delimiters of the staged range, one wave prefix sum;
//   * build_list:  the tile positions of the line starts of ranks [w0, w0 + kList] -> LDS;
//   * line k of the tile = [list[k], list[k + 1]) including its terminator; the last line's end may lie behind the staged range:
//     forward_scan reads on (rare: a line longer than `look`), and at the end of the stream an unterminated piece is a frame only
//     when the chunk is final (BufRead semantics);
//   * line_bad: any UTF-8 error bit inside the line.
// The rank of the tile's first line among all lines of the stream comes from a two-level look-back over per-tile counts (fg_fused.hpp,
// device only).  Everything HERE is written against fg_wave.hpp and runs lane for lane on the CPU emulation of a wave
// (tests/native/fuse_host.cpp, tests/test_fuse_cpu.py) against the oracle's restatement of the two splitters (fgo_frame).
#pragma once
#include "fg_fused_plan.hpp"
#include "fg_wave.hpp"

namespace fg {
namespace fuse {

constexpr uint32_t kPre = kPreBytes;  // bytes staged in front of the tile
constexpr uint32_t kList = 256;  // line starts listed at a time (list[0 .. kList] : kList lines and the end of the last)
constexpr uint32_t kUnresolved = 0xFFFFFFFFu;

// ---- byte classes of a 16-byte chunk ----------------------------------------------------------------------------------------------
// bit7 flags, per byte of a dword (fg_frame.hip's, spelled with the portable primitives)
FG_WV uint32_t f_cont(uint32_t b) { return b & ~(b << 1) & 0x80808080u; }  // 80..BF
FG_WV uint32_t f_ge_c0(uint32_t b) { return b & (b << 1) & 0x80808080u; }
FG_WV uint32_t f_ge_e0(uint32_t b) { return b & (b << 1) & (b << 2) & 0x80808080u; }
FG_WV uint32_t f_ge_f0(uint32_t b) { return b & (b << 1) & (b << 2) & (b << 3) & 0x80808080u; }
FG_WV uint32_t f_eq_hi(uint32_t b, uint32_t pat) {  // byte == pat's byte (pat bytes >= 0x80)
    const uint32_t y = b ^ pat;
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
}
// UTF-8 errors of the four bytes in `b`, given the dword before them: bit7 of byte i set <=> the well-formedness rules (Unicode 15
// Table 3-7) are violated AT byte i.  Judged from the byte and its three predecessors: no carried state, and no expectation ever
// crosses an ASCII byte -- such as a frame terminator --, so an error is always flagged inside the frame it belongs to.
FG_WV uint32_t utf8_err_flags(uint32_t b, uint32_t p) {
    const uint32_t p1 = wv::alignbyte(b, p, 3u);
    const uint32_t p2 = wv::alignbyte(b, p, 2u);
    const uint32_t p3 = wv::alignbyte(b, p, 1u);
    const uint32_t c1 = f_cont(p1), c2 = f_cont(p2);
    const uint32_t must = f_ge_c0(p1) | (f_ge_e0(p2) & c1) | (f_ge_f0(p3) & c2 & c1);
    uint32_t err = f_cont(b) ^ must;  // missing or stray continuation
    const uint32_t y = (b & 0xFEFEFEFEu) ^ 0xC0C0C0C0u;  // C0, C1
    err |= ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
    err |= b & ((b & 0x7F7F7F7Fu) + 0x0B0B0B0Bu) & 0x80808080u;  // >= F5
    const uint32_t b5 = b << 2, b4 = b << 3;  // second-byte ranges: E0 A0..BF | ED 80..9F | F0 90..BF | F4 80..8F
    err |= f_eq_hi(p1, 0xE0E0E0E0u) & ~b5;
    err |= f_eq_hi(p1, 0xEDEDEDEDu) & b5;
    err |= f_eq_hi(p1, 0xF0F0F0F0u) & ~b5 & ~b4;
    err |= f_eq_hi(p1, 0xF4F4F4F4u) & (b5 | b4);
    return err & 0x80808080u;
}
// delimiter mask (low 16 bits) | UTF-8 error mask (high 16 bits) of the chunk q0..q3 whose predecessor dword is pw.
//   rem = stream bytes from the chunk's first byte to the end of the stream, clamped to [-1, 16]: 16 = a full chunk, 0 .. 15 = the
//   chunk that holds position `nbytes` -- what lies behind the stream's end is the caller's memory, anything: it is read as zeros, so
//   that the error bit AT position nbytes says "a sequence cut off by the end of the stream" --, -1 = a chunk behind it (nothing).
FG_WV uint32_t chunk_masks(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t pw, uint32_t delim4, int32_t rem) {
    if (rem < 0) return 0u;
    if (rem < 16) {  // (rare: one chunk per stream)
        const uint32_t r = (uint32_t)rem;
        auto keep = [&](uint32_t d) -> uint32_t { return r >= 4u * d + 4u ? 0xFFFFFFFFu : r <= 4u * d ? 0u : (1u << (8u * (r - 4u * d))) - 1u; };
        q0 &= keep(0u), q1 &= keep(1u), q2 &= keep(2u), q3 &= keep(3u);
    }
    uint32_t dm = wv::gather16(wv::eq_flags(q0, delim4), wv::eq_flags(q1, delim4), wv::eq_flags(q2, delim4), wv::eq_flags(q3, delim4));
    uint32_t em = 0;
    if (((q0 | q1 | q2 | q3 | pw) & 0x80808080u) != 0u)
        em = wv::gather16(utf8_err_flags(q0, pw), utf8_err_flags(q1, q0), utf8_err_flags(q2, q1), utf8_err_flags(q3, q2));
    if (rem < 16) {
        dm &= (1u << rem) - 1u;
        em &= (2u << rem) - 1u;
    }
    return dm | (em << 16);
}

// ---- the tile ---------------------------------------------------------------------------------------------------------------------
// chunks a lane owns in the count / list passes: the tile's chunks dealt out in runs, a multiple of four (8-byte LDS reads)
FG_WVH uint32_t chunks_per_lane(uint32_t tile_cap) { return ((tile_cap / 16u + 63u) / 64u + 3u) & ~3u; }
// LDS of the fused part: dm16[64 * R] + list[kList + 2] (u16).  (The UTF-8 error masks are NOT kept: a tile without an error bit --
// every tile of a healthy stream -- never looks at them, and a tile with one re-derives them from the tile's own bytes, line_bad.  One
// word per chunk, both masks, cost the structured-data kernel a wave per CU.)
FG_WVH uint32_t lds_bytes(uint32_t tile_cap) { return 64u * chunks_per_lane(tile_cap) * 2u + (((kList + 2u) * 2u + 15u) & ~15u); }

struct Lds {
    uint16_t* dm16;  // [64 * R]: the delimiter mask of every 16-byte chunk of the staged range; zero behind the staged chunks
    uint16_t* list;  // [kList + 2]
    uint32_t R;
};
FG_WVH Lds carve(uint8_t* base, uint32_t tile_cap) {
    Lds L;
    L.R = chunks_per_lane(tile_cap);
    L.dm16 = reinterpret_cast<uint16_t*>(base);
    L.list = reinterpret_cast<uint16_t*>(base + 64u * L.R * 2u);
    return L;
}

// Geometry of tile T (wave-uniform).
struct Geo {
    uint64_t base;      // stream position of tile position 0 (= T * S - kPre; wraps below zero for T = 0: only ever ADDED to positions >= kPre)
    uint32_t span;      // staged bytes, a multiple of 16
    uint32_t own_end;   // delimiters at tile positions [kPre - 1, own_end) start a line of this tile
    uint32_t end_x;     // tile position of the stream's end (position `nbytes`) when the staged range reaches it, else kUnresolved
};
FG_WVH Geo tile_geo(uint64_t T, uint32_t S, uint32_t look, uint64_t nbytes) {
    Geo g;
    const uint64_t t0 = T * (uint64_t)S;
    g.base = t0 - kPre;
    const uint64_t left = nbytes - t0;  // (T < ntiles: >= 1)
    const uint32_t own = left < (uint64_t)S ? (uint32_t)left : S;
    g.own_end = kPre - 1u + own;
    if (left <= (uint64_t)S + look) {  // the stream ends inside the staged range: stage up to AND INCLUDING position nbytes
        g.end_x = kPre + (uint32_t)left;
        g.span = (g.end_x + 16u) & ~15u;
    } else {
        g.end_x = kUnresolved;
        g.span = kPre + S + look;  // (S and look are multiples of 16)
    }
    return g;
}
// rem argument of chunk_masks for tile chunk c (tile position 16 * c)
FG_WV int32_t chunk_rem(const Geo& g, uint32_t c) {
    if (g.end_x == kUnresolved) return 16;
    const int32_t r = (int32_t)g.end_x - (int32_t)(c * 16u);
    return r > 16 ? 16 : r < -1 ? -1 : r;
}
// the pre-chunk (tile chunk 0): only its LAST byte matters -- a terminator there starts a line at the tile's first byte; the stream's
// first tile has no byte before it and starts a line.  Errors in it belong to the tile before.
FG_WV uint32_t pre_chunk_mask(uint32_t m, bool first_tile) { return first_tile ? 0x8000u : (m & 0x8000u); }

struct Count {
    uint32_t n_own;  // lines that start in the tile
    uint32_t n_all;  // delimiters of the staged range (the pre-chunk's last byte included)
    uint32_t ex;     // this lane: delimiters before its run
};
// All 64 lanes.  The masks must be in LDS (and visible: wv::sync() before).
FG_WV Count count_tile(const Lds& L, const Geo& g) {
    const uint32_t lane = wv::lane();
    const uint32_t c0 = lane * L.R, x0 = c0 * 16u, x1 = x0 + L.R * 16u;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(L.dm16 + c0);  // two masks to the dword
    uint32_t all = 0;
    for (uint32_t j = 0; j < L.R / 2u; j += 2u) all += wv::popc32(w[j]) + wv::popc32(w[j + 1u]);
    // the delimiters in front of own_end: everything, nothing -- or, for the one lane whose run holds own_end, chunk by chunk
    uint32_t own = x1 <= g.own_end ? all : 0u;
    if (x0 < g.own_end && g.own_end < x1) {
        for (uint32_t j = 0; j < L.R; ++j) {
            const uint32_t c = x0 + j * 16u;
            if (c >= g.own_end) break;
            const uint32_t keep = c + 16u <= g.own_end ? 0xFFFFu : (1u << (g.own_end - c)) - 1u;
            own += wv::popc32((uint32_t)L.dm16[c0 + j] & keep);
        }
    }
    Count r;
    r.ex = wv::excl_sum(all, &r.n_all);
    const uint32_t lb = g.own_end / (L.R * 16u);  // the lane whose run holds own_end (beyond the last lane: every delimiter counts)
    r.n_own = wv::bcast(r.ex + own, lb < 63u ? lb : 63u);
    return r;
}
// list[r - w0] = tile position of the byte BEHIND the delimiter of rank r, for w0 <= r <= w0 + kList.  All 64 lanes; wv::sync() after.
FG_WV void build_list(const Lds& L, const Count& cn, uint32_t w0) {
    const uint32_t lane = wv::lane();
    const uint32_t c0 = lane * L.R;
    uint32_t r = cn.ex;
    if (r > w0 + kList) return;
    for (uint32_t j = 0; j < L.R; ++j) {
        uint32_t dm = L.dm16[c0 + j];
        while (dm) {
            const uint32_t b = wv::ctz32(dm);
            dm &= dm - 1u;
            if (r >= w0 && r <= w0 + kList) L.list[r - w0] = (uint16_t)((c0 + j) * 16u + b + 1u);
            ++r;
        }
        if (r > w0 + kList) break;
    }
}
// line k of the tile (w0 <= k < w0 + kList, k < n_own): tile positions [*s, *e) including the terminator; *e = kUnresolved when no
// delimiter follows inside the staged range (only ever the tile's last line).
FG_WV void line_at(const Lds& L, const Count& cn, uint32_t w0, uint32_t k, uint32_t* s, uint32_t* e) {
    *s = L.list[k - w0];
    *e = k + 1u < cn.n_all ? (uint32_t)L.list[k + 1u - w0] : kUnresolved;
}
// any UTF-8 error at tile positions [a, b) (a >= kPre, b <= span + 1)?  Re-derived from the tile's bytes (tile = the staged range as
// dwords): only a tile in which stage A saw an error bit ever asks.
FG_WV bool line_bad(const uint32_t* tile, const Geo& g, uint32_t a, uint32_t b) {
    for (uint32_t c = a >> 4; c * 16u < b; ++c) {
        const uint32_t* q = tile + c * 4u;
        uint32_t em = chunk_masks(q[0], q[1], q[2], q[3], q[-1], 0x01010101u, chunk_rem(g, c)) >> 16;
        const uint32_t x0 = c * 16u;
        if (x0 < a) em &= ~((1u << (a - x0)) - 1u);
        if (x0 + 16u > b) em &= (1u << (b - x0)) - 1u;
        if (em) return true;
    }
    return false;
}

// ---- the end of a line that runs past the staged range -------------------------------------------------------------------------------
// The wave reads on from stream position p0 (16-byte aligned), 1 KiB per step, until a delimiter or the end of the stream.
//   ld(pos) -> the 16 bytes at stream position pos as four dwords (zeros behind the readable range); pw0 = the dword before p0
// Returns the stream position BEHIND the delimiter, or ~0ull when the stream ends first; *bad |= a UTF-8 error in the bytes read up to
// there (up to and including position nbytes when the stream ends first).  All 64 lanes; the result is wave-uniform.
struct U4 { uint32_t x, y, z, w; };
template <class Load>
FG_WV uint64_t forward_scan(Load ld, uint64_t p0, uint64_t nbytes, uint32_t delim4, uint32_t pw0, bool* bad) {
    const uint32_t lane = wv::lane();
    uint32_t carry = pw0;
    bool any_bad = false;
    for (uint64_t pos = p0; pos <= nbytes; pos += 1024u) {
        const uint64_t cpos = pos + (uint64_t)lane * 16u;
        const U4 q = ld(cpos);
        const int32_t rem = cpos > nbytes ? -1 : nbytes - cpos >= 16u ? 16 : (int32_t)(nbytes - cpos);
        const uint32_t pw = wv::shfl_up1(q.w, carry);
        const uint32_t m = chunk_masks(q.x, q.y, q.z, q.w, pw, delim4, rem);
        const uint64_t hit = wv::ballot((m & 0xFFFFu) != 0u);
        if (hit) {
            const uint32_t l = wv::ctz64(hit);
            const uint32_t b = wv::ctz32(wv::shfl(m & 0xFFFFu, l));
            // errors in front of the delimiter: whole chunks of the lanes below l, the low bits of lane l's
            const uint32_t em = m >> 16;
            const bool e = lane < l ? em != 0u : lane == l ? (em & ((1u << b) - 1u)) != 0u : false;
            any_bad = any_bad || wv::any(e);
            *bad = *bad || any_bad;
            return pos + (uint64_t)l * 16u + b + 1u;
        }
        any_bad = any_bad || wv::any((m >> 16) != 0u);
        carry = wv::shfl(q.w, 63u);
    }
    *bad = *bad || any_bad;
    return ~0ull;
}

// The end of the tile's LAST line when no delimiter follows its start inside the staged range (cn->n_all == cn->n_own, cn->n_own != 0):
// behind the delimiter forward_scan finds -- or, when the stream ends first, the end of the stream for a FINAL chunk (BufRead: an
// unterminated last piece is a line) and no line at all otherwise (the caller carries the piece over: cn->n_own drops by one).
//   *tail_end  stream position behind the line;  *tail_bad  a UTF-8 error in the part of the line BEHIND the staged range
// All 64 lanes; wave-uniform results.
template <class Load>
FG_WV void resolve_tail(const Geo& g, Count* cn, Load ld, uint64_t nbytes, uint32_t delim4, bool final, uint32_t pw_last, uint64_t* tail_end,
                        bool* tail_bad) {
    *tail_end = 0;
    *tail_bad = false;
    if (cn->n_own == 0u || cn->n_all != cn->n_own) return;
    uint64_t e = ~0ull;
    if (g.end_x == kUnresolved) e = forward_scan(ld, g.base + g.span, nbytes, delim4, pw_last, tail_bad);
    if (e == ~0ull) {
        if (final) {
            e = nbytes;
        } else {
            cn->n_own -= 1u;
            *tail_bad = false;
        }
    }
    *tail_end = e;
}

}  // namespace fuse
}  // namespace fg
