// fg_sd_walk2.hpp -- the NEXT form of the lane-per-line structured-data walk of the RFC5424 kernel (host + device).
// reference: src/flowgger/decoder/rfc5424_decoder.rs:134-158 (parse_data's element loop) + :174-242 (parse_sd_data)
//
// STATUS: staged for round 4.  fg_rfc5424.hip still runs its own sd_walk_tile; this header has the same
// contract, is validated on the CPU against the oracle (tests/test_sd_walk2_cpu.py runs THIS source, compiled by g++, over the
// structured-data corpora and adversarial mutations of them -- the per-line walk lane by lane, the group step walk_group /
// copy_out on the fiber emulation of a wavefront) and has been compiled for gfx950 to count its instructions, but it has not
// run on a GPU yet and nothing in libfg_hip.so includes it.
//
// Why another form.  The measured cost of the corpus with ~13 pairs per line (profiles/r03z_cfg4_rfc5424_sd.json) is 306 VALU
// wave-instructions per line at 59 % VALU utilisation, and the pair loop is ~45 % of them: one iteration of sd_walk_tile is ~240
// VALU instructions and two to three DEPENDENT LDS round trips (bit scan of the quote bitmap from the value's first byte -> the
// closing quote p; a 16-byte window at p; five SWAR class masks of that window).  Here one iteration is ONE round trip and ~half
// the instructions:
//   * the 16-byte window at p and a 64-bit window of the quote/backslash bitmap at p are loaded TOGETHER (both addresses are known
//     at the top of the iteration).  The bytes give the next pair's prefix `" name="`, the bitmap window gives the next value's end
//     (the first quote/backslash bit behind the prefix) -- so the next iteration starts with its p in a register instead of a
//     bit scan through LDS.  Values longer than the window (rare in pairs, common in nothing) fall back to the scan.
//   * the common shape -- at most one space, a name of name characters, '=', '"', everything inside the window -- is decided
//     from ONE gathered mask ("is a name character": 33..=126 minus '=' ']', quotes and backslashes taken out with the bitmap
//     window) and two extracted bytes, instead of five masks.  Whatever does not fit (stray quotes, runs of spaces, a backslash
//     in a name, names of 14+ bytes, every error) takes the general step, which is sd_walk_tile's, statement for statement.
// The lean step accepts a window only when the general step would produce the same (name, value start) from it and no error, so
// the two forms are interchangeable per iteration -- the CPU test runs both (template parameter LEAN) and compares them.
#pragma once
#include <stdint.h>

#include "fg_tables_view.hpp"
#include "fg_wave.hpp"

// host builds count which step decided a pair prefix and where a value's end came from (the CPU test asserts that the corpus
// runs on the lean step with the bitmap window -- the premise of the instruction counts above)
#if defined(__HIPCC__)
#define FG_SD2_STAT(k) ((void)0)
#else
#define FG_SD2_STAT(k) (++::fg::sd2::stats()[k])
#endif

namespace fg {
namespace sd2 {
#if !defined(__HIPCC__)
enum { ST_LEAN = 0, ST_GENERAL = 1, ST_P_WINDOW = 2, ST_P_SCAN = 3, ST_N = 4 };
inline unsigned long long* stats() {
    static unsigned long long s[ST_N];
    return s;
}
#endif

// status codes == fg_rfc5424.hip's (index into the reference's error strings)
enum : uint32_t { E_OK = 0, E_NOMSG = 13, E_MALFORMED = 14, E_NOSD = 15, E_SDFMT = 16, E_NOBRACKET = 17 };
constexpr uint32_t E_HANDOFF = 0x100u;   // walk(): not an outcome of the line -- the first lane of a pair reached the second lane's start
constexpr uint32_t E_REDO = 0x101u;      // walk(): the first lane would have to read bytes behind a guess that did not hold (the second
                                         //         lane may have put its records there): the line is parsed again from a clean copy
constexpr uint32_t kNoSplit = 0xFFFFFFFFu;
enum { SD_COUNT = 0, SD_EMIT = 1, SD_STASH = 2 };

// The wave's tile as the walk sees it: bytes, and ONE bitmap with a bit per tile byte that is '"' or '\' (built by the whole
// wave for groups that hold structured data: rebuild_bitmap<QuoteClass>).  Both are readable 12 bytes / three bitmap dwords
// past the line's last byte (tile padding).
struct Tile {
    wv::Bytes b;
    const uint32_t* bm;
};

FG_WV bool is_name_char(uint32_t c) { return (c - 33u) <= 93u && c != '"' && c != '=' && c != ']'; }  // :188-192

// record: name_s | name_len << 16 | val_len << 32 | esc << 48 | is_sdid << 49   (val_s = name_s + name_len + 2)
FG_WV uint64_t pack(uint32_t name_s, uint32_t name_len, uint32_t val_len, uint32_t esc, uint32_t sdid) {
    return (uint64_t)name_s | ((uint64_t)name_len << 16) | ((uint64_t)val_len << 32) | ((uint64_t)esc << 48) | ((uint64_t)sdid << 49);
}

// bit7 of each byte set <=> 33 <= byte <= 126
FG_WV uint32_t range_flags(uint32_t x) {
    const uint32_t l = x & 0x7F7F7F7Fu;
    const uint32_t ge33 = l + 0x5F5F5F5Fu, is127 = l + 0x01010101u;
    return ge33 & ~is127 & ~x & 0x80808080u;
}
// bit7 of each byte set <=> the byte is in 33..=126 and is neither '=' nor ']'   (twelve operations per dword: the two equality
// tests share the masked operand and fold into the range test before the final mask)
FG_WV uint32_t name_flags(uint32_t x) {
    const uint32_t l = x & 0x7F7F7F7Fu;
    const uint32_t ge33 = l + 0x5F5F5F5Fu, is127 = l + 0x01010101u;
    const uint32_t ne_eq = (l ^ 0x3D3D3D3Du) + 0x7F7F7F7Fu;  // bit7 set <=> low 7 bits != '='
    const uint32_t ne_rb = (l ^ 0x5D5D5D5Du) + 0x7F7F7F7Fu;  //                           != ']'
    return ge33 & ne_eq & ne_rb & ~is127 & ~x & 0x80808080u;
}
// bytes k, k + 1 of a 16-byte window (k <= 14) in the low 16 bits.  The window comes BY VALUE: handed over as an array, the
// selects below become one load with a computed address and the window moves to scratch memory.
FG_WV uint32_t two_bytes(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t k) {
    const bool up = (k & 8u) != 0u, odd = (k & 4u) != 0u;
    const uint32_t a0 = up ? w2 : w0, a1 = up ? w3 : w1, a2 = up ? w3 : w2;  // (k >= 12: a2 is not looked at, k & 3 <= 2)
    return wv::alignbyte(odd ? a2 : a1, odd ? a1 : a0, k & 3u) & 0xFFFFu;
}

// 16-bit class masks of a 16-byte window for the general step (bit i <=> byte i)
struct WinMasks {
    uint32_t skip;  // ' ' or '"'   (the OUT state's ignorable characters, :194,:232)
    uint32_t name;  // 33..=126 minus '"' '=' ']'                                   :188-192
    uint32_t eq, quote, rb;
};
FG_WV WinMasks window_masks(const uint32_t w[4]) {
    uint32_t sp[4], qu[4], eq[4], rb[4], rg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sp[k] = wv::eq_flags(w[k], 0x20202020u);
        qu[k] = wv::eq_flags(w[k], 0x22222222u);
        eq[k] = wv::eq_flags(w[k], 0x3D3D3D3Du);
        rb[k] = wv::eq_flags(w[k], 0x5D5D5D5Du);
        rg[k] = range_flags(w[k]);
    }
    WinMasks m;
    m.quote = wv::gather16(qu[0], qu[1], qu[2], qu[3]);
    m.eq = wv::gather16(eq[0], eq[1], eq[2], eq[3]);
    m.rb = wv::gather16(rb[0], rb[1], rb[2], rb[3]);
    m.skip = wv::gather16(sp[0], sp[1], sp[2], sp[3]) | m.quote;
    m.name = wv::gather16(rg[0], rg[1], rg[2], rg[3]) & ~(m.quote | m.eq | m.rb);
    return m;
}

// Same contract as fg_rfc5424.hip's sd_walk_tile: pos = line index of the first '['; on success *msg_at = index of the ' ' that
// starts the message and *n_ent = entries (one per element id + one per pair).  MODE:
//   SD_COUNT  count only
//   SD_EMIT   write the entries to the table from `slot` on
//   SD_STASH  count AND leave an 8-byte record per entry in the line's own, already consumed tile bytes (from the line's first
//             4-byte boundary on); a record that would reach into bytes still to be read clears *rec_ok_out (the caller then
//             writes that line's entries by a walk over its copy in global memory)
//
// TWO LANES PER LINE (groups of at most 32 lines leave half the wave idle in stage B; the pair loop is a serial chain per line):
// the line's second lane starts in the middle of the structured data, at a byte it GUESSES to be the closing quote of a pair
// (pick_split), i.e. in the OUT state behind it (resume_i = that index + 1, pos is ignored); the first lane walks from the start
// with stop_at = that index and returns E_HANDOFF when it finds a pair closing exactly there -- then the second lane's start
// state was the true one and the line's result is the first lane's entries followed by the second lane's, with the second
// lane's status / msg_at.  If the first lane never closes a pair at stop_at (the guess was inside a value, in the message, ...)
// it cannot walk on in the tile -- the second lane has been leaving ITS records in the bytes behind the guess -- and returns
// E_REDO as soon as it would have to look at a byte behind stop_at, or reaches the end of the structured data before it (the caller parses that line again from global memory: the
// price of a wrong guess); an error it meets before that is the line's error (it comes first in the line).  Correctness never
// depends on the guess: when it holds, everything the first lane reads up to the hand-over lies at or before stop_at.
//   stop_at   kNoSplit, or the line index of the guessed closing quote
//   resume_i  0, or the line index to resume at in the OUT state (records then start at that byte's next 4-byte boundary)
template <int MODE, bool LEAN = true>
FG_WV uint32_t walk(const Tile& T, uint32_t base, uint32_t pos, uint32_t len, uint32_t* msg_at, uint32_t* n_ent, const DevTables& t,
                    uint32_t slot, uint32_t* tile_w = nullptr, bool* rec_ok_out = nullptr, uint32_t stop_at = kNoSplit,
                    uint32_t resume_i = 0u) {
    constexpr bool EMIT = MODE == SD_EMIT;
    uint32_t cnt = 0;
    uint32_t wpos = (base + resume_i + 3u) & ~3u;  // tile byte of the next record
    bool rec_ok = true;
    bool resume = resume_i != 0u;
    const bool first_of_two = stop_at != kNoSplit;
    // A record is written ONE entry late: entry k goes to the tile when entry k + 1 is complete (the last one when the walk ends), so
    // the bytes consumed since then are room as well.  The first lane of a line has its header's ~50 bytes ahead of its records; the
    // second lane starts with none, and without the delay a first pair shorter than 11 bytes (` k="ab"`) sent its line to the
    // walk over global memory -- 1.3 % of the corpus' lines, a third of its groups.
    uint64_t pend = 0;
    bool has_pend = false;
    auto flush = [&](uint32_t consumed) {  // consumed: line index up to which the walk is done with the bytes
        if (rec_ok && has_pend) {
            if (tile_w != nullptr && wpos + 8u <= base + consumed) {
                tile_w[wpos >> 2] = (uint32_t)pend;
                tile_w[(wpos >> 2) + 1u] = (uint32_t)(pend >> 32);
                wpos += 8u;
            } else {
                rec_ok = false;
            }
        }
        has_pend = false;
    };
    auto record = [&](uint64_t rec, uint32_t consumed) {
        flush(consumed);
        pend = rec;
        has_pend = true;
    };
    for (;;) {
        // sd_id = bytes after '[' up to the first ' ' (anything allowed)            :175-177
        const uint32_t s = pos + 1;
        uint32_t sp = resume ? resume_i - 1u : s;
        for (; !resume;) {  // 16 bytes per step
            if (sp >= len) return E_NOSD;
            if (first_of_two && sp + 15u > stop_at) return E_REDO;
            uint32_t w[4];
            T.b.load16(base + sp, w);
            const uint32_t avail = len - sp < 16u ? len - sp : 16u;
            const uint32_t hit = wv::gather16(wv::eq_flags(w[0], 0x20202020u), wv::eq_flags(w[1], 0x20202020u),
                                              wv::eq_flags(w[2], 0x20202020u), wv::eq_flags(w[3], 0x20202020u)) & ((1u << avail) - 1u);
            if (hit) {
                sp += wv::ctz32(hit);
                break;
            }
            sp += avail;
        }
        if (!resume) {
            if (EMIT) {
                t.ent_name[slot + cnt] = fg_span{s, sp - s};
                t.ent_val[slot + cnt] = 0;
                t.ent_type[slot + cnt] = FG_T_SDID;
                t.ent_flags[slot + cnt] = 0;
            }
            if (MODE == SD_STASH) record(pack(s, sp - s, 0, 0, 1), sp + 1u);
            ++cnt;
        }
        resume = false;
        uint32_t status = E_OK;
        uint32_t i = sp + 1;       // OUT state: next unread byte
        bool in_value = false;     // true: [val_s, ...) is an open value, `cur` = where to look for its end
        bool have_p = false;       // in_value: p_next is the first quote / backslash at or behind cur (no bit scan needed)
        uint32_t p_next = 0;
        uint32_t name_s = 0, name_e = 0, val_s = 0, cur = 0, esc_seen = 0;
        uint32_t close_at = 0;     // index of the element's ']'
        for (;;) {
            uint32_t w0 = i, start = 0;
            if (in_value) {
                uint32_t p = p_next;
                FG_SD2_STAT(have_p ? ST_P_WINDOW : ST_P_SCAN);
                if (!have_p) {
                    p = wv::find_bit<false>(T.bm, base + cur, base + len) - base;
                    if (cur >= len) p = len;
                }
                if (p >= len) {
                    status = E_NOBRACKET;  // input exhausted inside a value                  :239
                    break;
                }
                w0 = p;
                start = 1;
            } else if (i >= len) {
                status = E_NOBRACKET;
                break;
            }
            have_p = false;
            if (first_of_two && w0 > stop_at) return E_REDO;  // walked past the guess: no pair closes there
            // the window reaches behind the guess (at the guess itself, in a value, only byte 0 is looked at before the hand-over)
            const bool straddle = first_of_two && w0 + 15u > stop_at && !(in_value && w0 == stop_at);
            uint32_t w[4];
            T.b.load16(base + w0, w);
            const uint64_t qb = LEAN ? wv::window64(T.bm, base + w0) : 0ull;  // (issued with the window's bytes: one round trip)
            if (in_value) {
                if ((w[0] & 0xFFu) == '\\') {  // escapes the next char, whatever it is   :207-213
                    esc_seen = 1;
                    cur = w0 + 2;
                    if (LEAN) {  // the next quote / backslash behind the escaped byte, from the window when it shows one
                        const uint64_t m = qb >> 2;
                        p_next = w0 + 2u + wv::ctz64(m | (1ull << 63));
                        have_p = m != 0ull && p_next < len;
                    }
                    continue;
                }
                // closing quote: the pair is complete                                      :214-228
                if (EMIT) {
                    t.ent_name[slot + cnt] = fg_span{name_s, name_e - name_s};
                    t.ent_val[slot + cnt] = (uint64_t)val_s | ((uint64_t)(w0 - val_s) << 32);
                    t.ent_type[slot + cnt] = FG_T_STRING;
                    t.ent_flags[slot + cnt] = esc_seen ? FG_EF_VAL_ESC : 0;
                }
                if (MODE == SD_STASH) record(pack(name_s, name_e - name_s, w0 - val_s, esc_seen, 0), w0 + 1u);
                ++cnt;
                in_value = false;
                if (w0 == stop_at) {  // the second lane started right behind this quote, in the state the walk is in now
                    if (MODE == SD_STASH) flush(w0 + 1u);
                    *n_ent = cnt;
                    if (rec_ok_out) *rec_ok_out = rec_ok;
                    return E_HANDOFF;
                }
            }
            // ---- OUT state at window offset `start`: <skip chars> then ']' | name '=' '"' -----
            const uint32_t avail = len - w0 < 16u ? len - w0 : 16u;  // window bytes inside the line (>= 1)
            if (LEAN) {
                // the common shape, from one mask and two extracted bytes: [' '] name '=' '"' (or [' '] ']'), all inside the window
                const uint32_t nm = wv::gather16(name_flags(w[0]), name_flags(w[1]), name_flags(w[2]), name_flags(w[3])) & ~(uint32_t)qb & 0xFFFFu;
                const uint32_t b_start = start ? (w[0] >> 8) & 0xFFu : w[0] & 0xFFu;
                const uint32_t i0 = start + (b_start == ' ' ? 1u : 0u);          // at most one space is skipped here
                const uint32_t after = ~nm & ~((1u << i0) - 1u) & 0xFFFFu;         // first byte at / behind i0 that is not a name char
                const uint32_t e0 = after ? wv::ctz32(after) : 16u;
                if (e0 > i0 && e0 + 1u < avail) {                                 // (e0 <= 14: both bytes below are in the window)
                    if (straddle && w0 + e0 + 1u > stop_at) return E_REDO;
                    if (two_bytes(w[0], w[1], w[2], w[3], e0) == (('"' << 8) | '=')) {
                        FG_SD2_STAT(ST_LEAN);
                        name_s = w0 + i0;
                        name_e = w0 + e0;
                        val_s = name_e + 2u;
                        in_value = true;
                        esc_seen = 0;
                        cur = val_s;
                        const uint64_t m = qb >> (e0 + 2u);                       // quotes / backslashes from the value's first byte on
                        p_next = val_s + wv::ctz64(m | (1ull << 63));
                        have_p = m != 0ull && p_next < len;
                        continue;
                    }
                } else if (!straddle && e0 == i0 && i0 < avail && ((w[0] >> (8u * i0)) & 0xFFu) == ']') {  // (i0 <= 2)
                    close_at = w0 + i0;                                           // unescaped ']' outside name/value :197
                    break;
                }
                // anything else: the general step decides (same window)
            }
            if (straddle) return E_REDO;                               // (the general step looks at the whole window)
            const uint32_t inside = (1u << avail) - 1u;                // avail <= 16
            const WinMasks m = window_masks(w);
            const uint32_t from = ~((1u << start) - 1u);
            const uint32_t stop = ~m.skip & from & inside;             // first byte that is not ' ' / '"'
            if (stop == 0) {                                           // only ignorable bytes in view
                i = w0 + avail;
                continue;                                              // (i >= len is caught at the top)
            }
            const uint32_t i0 = wv::ctz32(stop);
            if ((m.rb >> i0) & 1u) {
                close_at = w0 + i0;                                    // unescaped ']' outside name/value :197
                break;
            }
            if (!((m.name >> i0) & 1u)) {
                status = E_SDFMT;                                      //                                   :235
                break;
            }
            // name = run of name chars from i0; then '=' and '"' must follow, all inside the view
            const uint32_t after_name = ~m.name & ~((1u << i0) - 1u) & 0xFFFFu;
            const uint32_t e0 = after_name ? wv::ctz32(after_name) : 16u;
            if (e0 + 1u < avail) {
                if (!((m.eq >> e0) & 1u) || !((m.quote >> (e0 + 1u)) & 1u)) {
                    status = E_SDFMT;
                    break;
                }
                name_s = w0 + i0;
                name_e = w0 + e0;
                val_s = name_e + 2u;
            } else if (i0 != 0) {
                i = w0 + i0;                                           // re-window with the name at offset 0
                continue;
            } else {
                // a name of 14+ bytes (or the line ends inside this prefix): byte-wise
                uint32_t q = w0, c = 0;
                do {
                    ++q;
                    if (first_of_two && q + 1u > stop_at) return E_REDO;
                    c = q < len ? T.b.byte(base + q) : 0x100u;
                } while (is_name_char(c));
                if (q >= len || q + 1u >= len) {
                    // exhausted in IN_NAME / HAVE_NAME (a non-'=' / non-'"' byte there is a format error first)
                    status = (q < len && c != '=') ? E_SDFMT : E_NOBRACKET;
                    break;
                }
                if (c != '=' || T.b.byte(base + q + 1u) != '"') {
                    status = E_SDFMT;
                    break;
                }
                name_s = w0;
                name_e = q;
                val_s = q + 2u;
            }
            FG_SD2_STAT(ST_GENERAL);
            in_value = true;
            esc_seen = 0;
            cur = val_s;
        }
        if (status != E_OK) return status;
        const uint32_t after = close_at + 1;
        if (first_of_two && after > stop_at) return E_REDO;
        if (after >= len) return E_NOMSG;  // :148
        const uint32_t c = T.b.byte(base + after);
        if (c == '[') {
            pos = after;
            continue;
        }
        if (c != ' ') return E_MALFORMED;  // :154
        // the structured data ends before the guess (which then lies in the message): the walk's result would be right, but the
        // caller's trims of the message would read bytes the second lane may have written to
        if (first_of_two) return E_REDO;
        if (MODE == SD_STASH) flush(after);  // (the bytes from `after` on are the caller's: the message)
        *msg_at = after;
        *n_ent = cnt;
        if (rec_ok_out) *rec_ok_out = rec_ok;
        return E_OK;
    }
}


// Where the second lane of a line starts: a byte that looks like the closing quote of a pair near the MIDDLE of the structured
// data.  The structured data's extent is not known before it is walked, so the middle is estimated from the bitmap: quotes are
// dense in structured data and rare in message text, hence the LAST quote / backslash bit of the line (looked for in its final
// 480 bytes) is taken for the end, and the first bit at or behind the midpoint between `pos` and that end which looks like a
// closing quote -- a '"' that is not escaped, does not follow '=' and is followed by ' ' or ']' -- is the guess (at most four
// candidates are looked at).  An unescaped quote inside a value does not exist, so inside the structured data such a byte IS a
// closing quote; a guess in a message that has quotes of its own is the case the first lane answers with E_REDO.  Returns the
// line index of the guess, or kNoSplit (less than 128 bytes between `pos` and the last bit: not worth a second lane; no bit in
// the final 480 bytes: a long message; nothing plausible near the middle).
FG_WV uint32_t pick_split(const Tile& T, uint32_t base, uint32_t pos, uint32_t len) {
    // ---- the last set bit of the line: sixteen bitmap words that end at the line's end, all in flight, highest non-zero wins
    const uint32_t endb = base + len;                     // one past the line's last tile bit
    const uint32_t d1 = (endb + 31u) >> 5;                // word index one past the word that holds bit endb - 1
    uint32_t x[16];
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) x[k] = T.bm[d1 >= k + 1u ? d1 - 1u - k : 0u];  // (k = 0: the word with the line's end)
    if ((endb & 31u) != 0u) x[0] &= (1u << (endb & 31u)) - 1u;                        // bits of the next line
    uint32_t top = 0u, top_k = 16u;
#pragma unroll
    for (int k = 15; k >= 0; --k) {  // the lowest k with a set bit wins (selects, no branches: the loads above stay in flight together)
        const bool hit = x[k] != 0u && d1 >= (uint32_t)k + 1u;
        top = hit ? x[k] : top;
        top_k = hit ? (uint32_t)k : top_k;
    }
    if (top_k == 16u) return kNoSplit;
    const uint32_t last_bit = 32u * (d1 - 1u - top_k) + 31u - (uint32_t)wv::clz64((uint64_t)top << 32);
    if (last_bit < base + pos + 128u) return kNoSplit;    // (also: the bit lies before `pos`, e.g. a quote in the header)
    const uint32_t last = last_bit - base;
    // ---- candidates from the midpoint on
    uint32_t res = wv::find_bit<false>(T.bm, base + pos + ((last - pos) >> 1), base + len) - base;
    for (uint32_t tries = 0; tries < 4u; ++tries) {
        if (res + 1u >= len) return kNoSplit;
        const uint64_t v = T.b.load8(base + res - 1u);  // (res > pos >= 1)
        const uint32_t prev = (uint32_t)v & 0xFFu, at = (uint32_t)(v >> 8) & 0xFFu, next = (uint32_t)(v >> 16) & 0xFFu;
        if (at == '"' && prev != '\\' && prev != '=' && (next == ' ' || next == ']')) return res;
        res = wv::find_bit<false>(T.bm, base + res + 1u, base + len) - base;
    }
    return kNoSplit;
}

// ---------------------------------------------------------------------------------------------
// The walk of a whole group, wave-cooperative (all 64 lanes call it in wave-uniform control flow): what stage B of the RFC5424
// kernel does between the header fast path and the entry slots.
//   two    wave-uniform: the group holds at most 32 lines (lanes 32..63 have none) -- each line with structured data gets lane
//          l + 32 as its second lane
//   sd     this lane's line has structured data to walk at line index `pos` (line = tile bytes [base, base + len))
// ---------------------------------------------------------------------------------------------
struct PairOut {
    // the line's lane:
    uint32_t status;  // E_OK or the walk's error (meaningless when redo)
    uint32_t n_ent;   // entries of the line (0 unless E_OK)
    uint32_t msg_at;  // index of the ' ' that starts the message (E_OK)
    bool rec_ok;      // every entry of the line sits in a record in the tile (else: EMIT walk over a clean copy of the line)
    bool redo;        // the second lane's guess did not hold: parse the line again from a clean copy (its tile bytes are not intact)
    // both lanes -- the copy-out: this lane holds n_own records from tile byte rec_at on; they are entries [skip, skip + n_own) of
    // the line of lane l & 31 (n_own = 0 whenever the line is not E_OK / rec_ok / redo)
    uint32_t n_own, rec_at, skip;
};
template <int MODE>
FG_WV PairOut walk_group(const Tile& T, uint32_t* tile_w, bool two, bool sd, uint32_t base, uint32_t pos, uint32_t len, const DevTables& t) {
    PairOut o{E_OK, 0u, 0u, false, false, 0u, 0u, 0u};
    const uint32_t l = wv::lane();
    const bool second = two && l >= 32u;
    // the second lane adopts the line of lane l - 32  (ONE inlined walk for every case: the walk is the kernel's largest piece of
    // code, and a copy per mode was measurably worse than the instructions it saved)
    const uint32_t src = two ? (l & 31u) : l;
    const bool sdl = wv::shfl(sd ? 1u : 0u, src) != 0u;
    const uint32_t b = wv::shfl(base, src), p = wv::shfl(pos, src), n = wv::shfl(len, src);
    const uint32_t split = (two && sdl) ? pick_split(T, b, p, n) : kNoSplit;  // (the same value in both lanes of a line)
    uint32_t st = E_OK, cnt = 0, m = 0;
    bool ok = false;
    if (sdl && (!second || split != kNoSplit))
        st = walk<MODE>(T, b, p, n, &m, &cnt, t, 0u, tile_w, &ok, second ? kNoSplit : split, second ? split + 1u : 0u);
    if (!two) {
        if (sdl && st == E_OK) {
            o.n_ent = cnt;
            o.msg_at = m;
            o.rec_ok = ok;
            o.n_own = ok ? cnt : 0u;
            o.rec_at = (b + 3u) & ~3u;
        }
        o.status = sdl ? st : E_OK;
        return o;
    }
    // exchange across the halves
    const uint32_t st_b = wv::shfl(st, l | 32u), cnt_b = wv::shfl(cnt, l | 32u), m_b = wv::shfl(m, l | 32u), ok_b = wv::shfl(ok ? 1u : 0u, l | 32u);
    const uint32_t st_a = wv::shfl(st, l & 31u), cnt_a = wv::shfl(cnt, l & 31u), ok_a = wv::shfl(ok ? 1u : 0u, l & 31u);
    if (!sdl) return o;
    const bool handed = split != kNoSplit && st_a == E_HANDOFF;
    const bool line_ok = handed ? st_b == E_OK : st_a == E_OK;
    const bool recs = line_ok && ok_a != 0u && (!handed || ok_b != 0u);
    if (!second) {
        if (handed) {
            o.status = st_b;
            if (st_b == E_OK) {
                o.n_ent = cnt + cnt_b;
                o.msg_at = m_b;
            }
        } else if (st == E_REDO) {
            o.redo = true;
        } else {
            o.status = st;  // the whole walk (no second lane was started), or an error before the split
            if (st == E_OK) {
                o.n_ent = cnt;
                o.msg_at = m;
            }
        }
        o.rec_ok = recs;
        o.n_own = recs ? cnt : 0u;
        o.rec_at = (b + 3u) & ~3u;
    } else {
        o.n_own = (handed && recs) ? cnt : 0u;
        o.rec_at = (b + split + 1u + 3u) & ~3u;
        o.skip = cnt_a;
    }
    return o;
}

// The records of one lane -> entry slots first .. first + n (four records in flight before the first store)
FG_WV void copy_out(const uint32_t* tile_w, uint32_t rec_at, uint32_t n, uint32_t first, const DevTables& t) {
    const uint32_t* rec32 = tile_w + (rec_at >> 2);
    for (uint32_t k0 = 0; k0 < n; k0 += 4u) {
        uint32_t lo[4], hi[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t k = k0 + j < n ? k0 + j : n - 1u;
            lo[j] = rec32[k * 2u];
            hi[j] = rec32[k * 2u + 1u];
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t k = k0 + j;
            if (k < n) {
                const uint32_t name_s = lo[j] & 0xFFFFu, name_len = lo[j] >> 16, val_len = hi[j] & 0xFFFFu;
                const bool sdid = (hi[j] >> 17) & 1u;
                t.ent_name[first + k] = fg_span{name_s, name_len};
                t.ent_val[first + k] = sdid ? 0ull : ((uint64_t)(name_s + name_len + 2u) | ((uint64_t)val_len << 32));
                t.ent_type[first + k] = sdid ? FG_T_SDID : FG_T_STRING;
                t.ent_flags[first + k] = ((hi[j] >> 16) & 1u) ? FG_EF_VAL_ESC : 0;
            }
        }
    }
}

}  // namespace sd2
}  // namespace fg
