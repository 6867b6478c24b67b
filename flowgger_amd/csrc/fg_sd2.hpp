// fg_sd2.hpp -- wave-cooperative, PAIR-PARALLEL structured-data walk for RFC5424Decoder::decode
// (reference: src/flowgger/decoder/rfc5424_decoder.rs:127-242 -- parse_data, parse_sd_data, the 6-tuple state machine at :187-237).
//
// The lane-per-line walker of fg_rfc5424.hip costs a group max-over-lanes of the pair count (16-18 trips for a mean of 13) with
// half of the wave's lanes holding a line, and every trip is a chain of dependent LDS round trips in ONE lane: 306 VALU
// wave-instructions per 554-byte line, 1.2 G lines/s (profiles/r04a_ab_pmc_main.json).  Here the work item is a PAIR:
//
//   chunk pass  (lane = 16 tile bytes)  quote and backslash masks by SWAR; escaped characters by the odd-backslash-run arithmetic
//                                       (16-bit form, carry between neighbouring lanes) -> bitmap of REAL quotes + backslash bitmap
//   word pass   (lane = 64-byte words)  quote counts per word (wave prefix sum), positions of all real quotes -> items[]
//   line pass   (lane = line)           the line's first quote at / behind its '[', its pair slots (two quotes each)
//   P1          (lane = pair slot)      every slot of the tile, 64 per trip: the everyday pair ` name="value"` is PROVEN from one
//                                       16-byte window in front of the opening quote (name-character mask by SWAR, '=' and the
//                                       single space by byte), the closing `"] ` is the line's terminal; anything else goes to a list
//   P2          (lane = listed slot)    the EXACT state machine of the reference over the bytes between the previous closing quote
//                                       and this opening quote (element starts `[id `, `][id `, spaces, errors): every error message
//                                       of the reference is decided here, by the first event in line order
//   verdict     (lane = line)           first event of the line (LDS min): terminal -> Ok with J pairs; error -> the exact status;
//                                       no event among the slots -> the same state machine over the tail behind the last pair
//   emit        (lane = pair slot)      entries straight to the table at first(line) + pair index + elements started so far
//
// Exactness: a FAST FORM.  A line is handled only when every byte of its structured data has been proven to be part of the shape
// above (or an error has been proven to be the FIRST one the reference would raise); everything else -- stray quotes, more than
// one space between an sd_id and its first name, empty elements without an error, names over 62 bytes, gaps over 64 bytes, more
// than 64 pairs, 16 backslashes in a row ... -- is handed back untouched (`handled = false`) and takes the byte-wise walker.
//
// Portable: compiled by hipcc for gfx950 and by g++ over the fiber emulation of a wave (fg_wave.hpp) for the CPU suite.
#pragma once
#include "fg_tables_view.hpp"
#include "fg_wave.hpp"

namespace fg {
namespace sd2 {

// status codes == fg_rfc5424.hip's (index into the reference's error strings)
enum : uint32_t { E_OK = 0, E_NOMSG = 13, E_MALFORMED = 14, E_NOSD = 15, E_SDFMT = 16, E_NOBRACKET = 17 };

constexpr uint32_t kMaxPairs = 64;   // pair slots tracked per line (the element mask is 64 bits wide)
constexpr uint32_t kGapBudget = 64;  // bytes the exact state machine walks per gap / tail before it gives the line back
constexpr uint32_t kNone = 0xFFFFFFFFu;

// ---------------------------------------------------------------------------------------------
// LDS beyond the tile and its two bitmaps (real quotes, backslashes): the "extra" block
// ---------------------------------------------------------------------------------------------
struct Lds {
    wv::Bytes T;        // tile bytes
    uint32_t* tile_w;   // ... writable (the lines' info blocks live in their own, already parsed header bytes)
    uint32_t* bmQ;      // REAL quotes, one bit per tile byte (readable three dwords past the tile's last bit word)
    uint32_t* bmB;      // backslashes
    uint16_t* items;    // [item_cap + 2] tile position of every real quote; items[-1] exists (one pad entry in front)
    uint16_t* qcnt;     // [words + 2] real quotes before each 64-byte word
    uint16_t* rec;      // [item_cap / 2 + 4] per pair slot: name_len | esc << 6 | elem << 7 | kind << 8 | payload << 10
    uint16_t* owner;    // [item_cap / 2 + 64] per pair slot: tile offset of its line's info block (built once per tile: no
                        // barrier inside the slot loops)
    uint16_t* slow;     // [slow_cap] P2 list: pair slot (tile-wide)
    uint32_t item_cap, slow_cap;
};
// info block of a line, six dwords at the line's first 4-byte boundary (the header bytes are dead once the fast path has the row)
enum { I_TV = 0, I_EL0 = 1, I_EL1 = 2, I_FIRST = 3, I_I0CUM = 4, I_S0E = 5, I_BASEW = 6, kInfoWords = 7 };  // (28 bytes: d0 >= 32)
enum : uint32_t { K_PAIR = 0, K_TERM = 1, K_ERR = 2, K_BAIL = 3 };

FG_WVH uint32_t up8(uint32_t v) { return (v + 7u) & ~7u; }
FG_WVH uint32_t item_cap_for(uint32_t tile_cap) { return tile_cap / 16u; }  // one real quote per 16 bytes (corpus: one per 21)
FG_WVH uint32_t slow_cap_for(uint32_t tile_cap) { return tile_cap / 64u < 128u ? 128u : tile_cap / 64u; }
FG_WVH uint32_t extra_bytes(uint32_t tile_cap) {
    const uint32_t words = tile_cap / 64u + 2u, ic = item_cap_for(tile_cap), sc = slow_cap_for(tile_cap);
    return up8((ic + 4u) * 2u) + up8(words * 2u) + up8((ic / 2u + 4u) * 2u) + up8((ic / 2u + 64u) * 2u) + up8(sc * 2u);
}
FG_WV Lds carve(const uint8_t* tile, uint16_t* bm16, uint32_t tile_cap, uint8_t* extra) {
    Lds L;
    L.T.w = reinterpret_cast<const uint32_t*>(tile);
    L.tile_w = reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(tile));
    const uint32_t stride16 = tile_cap / 16u + 16u;
    L.bmQ = reinterpret_cast<uint32_t*>(bm16);
    L.bmB = reinterpret_cast<uint32_t*>(bm16 + stride16);
    L.item_cap = item_cap_for(tile_cap);
    L.slow_cap = slow_cap_for(tile_cap);
    const uint32_t words = tile_cap / 64u + 2u;
    uint8_t* p = extra;
    L.items = reinterpret_cast<uint16_t*>(p) + 1; p += up8((L.item_cap + 4u) * 2u);
    L.qcnt = reinterpret_cast<uint16_t*>(p); p += up8(words * 2u);
    L.rec = reinterpret_cast<uint16_t*>(p); p += up8((L.item_cap / 2u + 4u) * 2u);
    L.slow = reinterpret_cast<uint16_t*>(p); p += up8(L.slow_cap * 2u);
    L.owner = reinterpret_cast<uint16_t*>(p);
    return L;
}

FG_WV uint64_t below(uint32_t bit) { return bit >= 64u ? ~0ull : (1ull << bit) - 1ull; }

// characters escaped by a backslash inside a 16-byte chunk (simdjson's odd-backslash-run arithmetic on 16-bit masks);
// cin = byte 0 is escaped by the chunk before
FG_WV uint32_t find_escaped16(uint32_t bs, uint32_t cin) {
    bs &= ~cin;
    const uint32_t follows = ((bs << 1) | cin) & 0xFFFFu;
    const uint32_t even = 0x5555u;
    const uint32_t odd_starts = bs & ~even & ~follows;
    const uint32_t seq_even = (odd_starts + bs) & 0xFFFFu;
    const uint32_t invert = (seq_even << 1) & 0xFFFFu;
    return (even ^ invert) & follows;
}
// bit 7 of each byte set <=> the byte is an SD-NAME character: 33..=126 minus '"' '=' ']'   (rfc5424_decoder.rs:188-192)
FG_WV uint32_t name_flags(uint32_t x) {
    const uint32_t t = x & 0x7F7F7F7Fu;
    const uint32_t ge33 = t + 0x5F5F5F5Fu;                      // bit7 <=> low7 >= 33
    const uint32_t is127 = t + 0x01010101u;                     // bit7 <=> low7 == 127
    const uint32_t nq = (t ^ 0x22222222u) + 0x7F7F7F7Fu;        // bit7 <=> low7 != '"'
    const uint32_t ne = (t ^ 0x3D3D3D3Du) + 0x7F7F7F7Fu;        // ... != '='
    const uint32_t nb = (t ^ 0x5D5D5D5Du) + 0x7F7F7F7Fu;        // ... != ']'
    return ge33 & ~is127 & nq & ne & nb & ~x & 0x80808080u;
}
FG_WV bool is_name_char(uint32_t c) { return (c - 33u) <= 93u && c != '"' && c != '=' && c != ']'; }

// ---------------------------------------------------------------------------------------------
// chunk pass: the two bitmaps of the staged tile.  Returns true when a chunk of sixteen backslashes was met (the escape carry
// would chain through it: the whole tile takes the byte-wise walker).  All 64 lanes, wave-uniform.
// ---------------------------------------------------------------------------------------------
FG_WV bool classify_tile(const Lds& L, uint32_t span) {
    const uint32_t lane = wv::lane();
    const uint32_t nchunk = span >> 4;
    uint16_t* q16 = reinterpret_cast<uint16_t*>(L.bmQ);
    uint16_t* b16 = reinterpret_cast<uint16_t*>(L.bmB);
    bool chain = false;
    uint32_t carry = 0;  // the last chunk of the row before ends in an odd run of backslashes (wave-uniform)
    for (uint32_t c0 = 0; c0 < nchunk; c0 += wv::kLanes) {
        const uint32_t c = c0 + lane;
        const bool in = c < nchunk;
        const uint32_t a = (in ? c : nchunk - 1u) * 16u;
        uint32_t x[4];
        L.T.load16(a, x);  // (16-byte aligned: four plain dword reads)
        uint32_t q = wv::gather16(wv::eq_flags(x[0], 0x22222222u), wv::eq_flags(x[1], 0x22222222u), wv::eq_flags(x[2], 0x22222222u),
                                  wv::eq_flags(x[3], 0x22222222u));
        uint32_t b = wv::gather16(wv::eq_flags(x[0], 0x5C5C5C5Cu), wv::eq_flags(x[1], 0x5C5C5C5Cu), wv::eq_flags(x[2], 0x5C5C5C5Cu),
                                  wv::eq_flags(x[3], 0x5C5C5C5Cu));
        if (!in) q = b = 0u;
        chain = chain || b == 0xFFFFu;
        // parity of the run of backslashes that ends at the chunk's last byte (0 when the last byte is none)
        const uint32_t run = wv::clz64(~((uint64_t)b << 48));  // leading ones of the 16-bit field
        const uint32_t odd = run & 1u;
        const uint32_t cin = wv::shfl_up1(odd, carry);
        carry = wv::bcast(odd, 63u);
        const uint32_t qr = q & ~find_escaped16(b, cin);
        if (in) {
            q16[c] = (uint16_t)qr;
            b16[c] = (uint16_t)b;
        }
    }
    // zeros behind the last chunk: the 64-bit windows read up to three dwords past it
    for (uint32_t c = nchunk + lane; c < nchunk + 8u; c += wv::kLanes) {
        q16[c] = 0;
        b16[c] = 0;
    }
    return wv::any(chain);
}

// ---------------------------------------------------------------------------------------------
// The reference's state machine over a stretch of the line, byte by byte (rfc5424_decoder.rs:174-242 + parse_data :134-158).
//   from      tile position of the first byte to look at
//   at_elem   true: `from` is the '[' that opens an element (the line's first); false: the OUT state inside an element
//   open      tile position of the next opening quote (the walk must arrive there in the HAVE_NAME state), or kNone: a TAIL -- the
//             walk runs to the end of the line
//   end       end of the line's bytes in the tile; line_end_known = false when the line continues behind `end` (HEAD staging)
// What comes back is the FIRST event in byte order.
// ---------------------------------------------------------------------------------------------
struct Event {
    uint32_t kind;       // K_PAIR: arrived at `open` as the reference would (name_len etc. valid); K_TERM; K_ERR; K_BAIL
    uint32_t status;     // K_ERR: the reference's error
    uint32_t msg_at;     // K_TERM: tile position of the ' ' that starts the message
    uint32_t name_len;   // K_PAIR
    uint32_t elem;       // K_PAIR: 1 = exactly one element starts in this stretch, in the canonical spelling `[id name=` / `][id name=`
};
FG_WV Event gap_walk(const wv::Bytes& T, uint32_t from, bool at_elem, uint32_t open, uint32_t end, bool line_end_known) {
    Event ev{K_BAIL, E_OK, 0u, 0u, 0u};
    const uint32_t full = open != kNone ? open + 1u : end;  // bytes [from, full) are to be walked (the opening quote included)
    const uint32_t lim = full - from > kGapBudget ? from + kGapBudget : full;
    // states: 0 OUT  1 IN_NAME  2 HAVE_NAME  5 SDID (behind a '[' up to the first ' ')  6 AFTER (behind an element's ']')
    uint32_t st = at_elem ? 6u : 0u, name_s = 0, elems = 0, spaces = 0;
    bool canon = true;  // one element at most, opened as the stretch's first bytes, ONE space between its id and the name
    uint32_t i = from;
    // 16 bytes per LDS round trip
    uint32_t w0 = kNone;
    uint64_t lo = 0, hi = 0;
    auto byte_at = [&](uint32_t p) -> uint32_t {
        uint32_t off = p - w0;
        if (w0 == kNone || off >= 16u) {
            uint32_t q[4];
            T.load16(p, q);
            lo = (uint64_t)q[0] | ((uint64_t)q[1] << 32);
            hi = (uint64_t)q[2] | ((uint64_t)q[3] << 32);
            w0 = p;
            off = 0;
        }
        const uint64_t v = off < 8u ? lo : hi;
        return (uint32_t)(v >> (8u * (off & 7u))) & 0xFFu;
    };
    for (; i < lim; ++i) {
        const uint32_t c = byte_at(i);
        if (st == 6u) {  // parse_data :134-158: what follows an element (or: the line's first '[')
            if (c == '[') {
                st = 5u;
                ++elems;
                if (i != from + (at_elem ? 0u : 1u)) canon = false;  // `[` first (line's first element) or right behind `]`
            } else if (c == ' ') {
                if (elems) return ev;  // an element without pairs opened in this stretch (`[id ] msg`): its sd_id entry has no pair lane
                ev.kind = K_TERM;
                ev.msg_at = i;
                return ev;
            } else {
                ev.kind = K_ERR;
                ev.status = E_MALFORMED;  // :154
                return ev;
            }
        } else if (st == 5u) {  // sd_id: anything up to the first ' '   :175-177
            if (c == ' ') {
                st = 0u;
                spaces = 1;
            } else if (open != kNone && i == open) {
                return ev;  // a quote inside an sd_id: the pairing of the quotes is off -- not fast-form material
            }
        } else if (st == 0u) {
            if (c == ' ') {
                ++spaces;
            } else if (c == '"') {
                return ev;  // tolerated stray quote (:232): the pairing of the quotes is off
            } else if (c == ']') {
                st = 6u;  // :197
                if (elems) canon = false;  // an element that opened in this stretch closes in it: no pairs
            } else if (is_name_char(c)) {
                st = 1u;
                name_s = i;
            } else {
                ev.kind = K_ERR;
                ev.status = E_SDFMT;  // :235
                return ev;
            }
        } else if (st == 1u) {
            if (is_name_char(c)) {
            } else if (c == '=') {
                st = 2u;
                ev.name_len = i - name_s;
            } else {
                ev.kind = K_ERR;
                ev.status = E_SDFMT;
                return ev;
            }
        } else {  // st == 2: HAVE_NAME expects the opening quote
            if (c != '"') {
                ev.kind = K_ERR;
                ev.status = E_SDFMT;
                return ev;
            }
            if (open == kNone || i != open) return ev;  // (cannot happen: every real quote is an item)
            // arrived: the canonical spellings are ` name=` (no element) and `[id name=` / `][id name=`
            const bool one_space = elems ? spaces == 1u : spaces <= 1u;
            if (elems > 1u || !canon || !one_space || ev.name_len > 62u) return ev;
            ev.kind = K_PAIR;
            ev.elem = elems;
            return ev;
        }
    }
    if (lim != full) return ev;  // the budget is used up without an event
    // the stretch is used up
    if (open != kNone) return ev;  // did not arrive at the opening quote in HAVE_NAME (escaped quotes in between ...)
    if (!line_end_known) return ev;
    // a TAIL ran into the end of the line
    ev.kind = K_ERR;
    if (st == 6u) ev.status = E_NOMSG;            // :148  (an element just closed: nothing follows)
    else if (st == 5u) ev.status = E_NOSD;        // :177  no space behind the sd_id
    else ev.status = E_NOBRACKET;                 // :239  input exhausted inside an element
    return ev;
}

// The stretch in front of an opening quote that OPENS AN ELEMENT in the canonical spelling -- `[id name=` (the line's first stretch,
// from the '[' on) or `][id name=` (behind a closing quote) -- proven from byte-class masks of the 48 bytes before the quote instead
// of the byte-wise walk (a serial walk of ~18 bytes cost the group a quarter of its time: profiles/r04f_phases_cfg4_pairs_v2.log):
// the id is whatever stands before the stretch's ONLY space (rfc5424_decoder.rs:175-177: anything but a space), the name a run of
// name characters up to the '='.  Returns the name's length, 0 = not that shape (the exact walk decides).  gs = the stretch's first
// byte, first = it is the line's first stretch.
FG_WV uint32_t elem_start_shape(const wv::Bytes& T, uint32_t gs, uint32_t open, bool first) {
    const uint32_t g = open - gs;
    if (g > 48u || g < 4u || open < 48u) return 0u;
    uint64_t nm = 0, sp = 0;
    uint32_t last = 0;
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) {
        uint32_t w[4];
        T.load16(open - 48u + 16u * k, w);
        nm |= (uint64_t)wv::gather16(name_flags(w[0]), name_flags(w[1]), name_flags(w[2]), name_flags(w[3])) << (16u * k);
        sp |= (uint64_t)wv::gather16(wv::eq_flags(w[0], 0x20202020u), wv::eq_flags(w[1], 0x20202020u), wv::eq_flags(w[2], 0x20202020u),
                                     wv::eq_flags(w[3], 0x20202020u)) << (16u * k);
        last = w[3];
    }
    // bit i <=> byte open - 48 + i; the stretch = bits [48 - g, 48); byte 47 must be the '='
    const uint64_t not_nm = (~nm << 17) | (1ull << 16);      // bit 63 <=> byte 46 is no name character
    const uint32_t nl = wv::clz64(not_nm);                   // name characters that end at byte 46 (at most 47)
    const uint32_t pre = first ? 1u : 2u;                    // `[` / `][`
    if ((last >> 24) != '=' || nl == 0u || nl > 62u || nl + 2u + pre > g) return 0u;
    const uint32_t spb = 46u - nl;                           // the byte in front of the name: the ONE space of the stretch
    const uint64_t in = (~0ull << (48u - g)) & ((1ull << 48) - 1ull);
    if ((sp & in) != (1ull << spb)) return 0u;
    const uint64_t head = T.load8(gs);
    const uint32_t b0 = (uint32_t)head & 0xFFu, b1 = (uint32_t)(head >> 8) & 0xFFu;
    if (first ? b0 != '[' : (b0 != ']' || b1 != '[')) return 0u;
    return nl;
}

// ---------------------------------------------------------------------------------------------
// What the lane that owns a line hands in and gets back
// ---------------------------------------------------------------------------------------------
struct LineIn {
    bool sd;          // this lane's line has structured data to walk ('[' at d0) and lies in the tile from its first byte
    uint32_t base;    // tile position of the line's first byte
    uint32_t d0;      // line index of the '['
    uint32_t wlen;    // bytes of the line that are in the tile (== its length unless HEAD staging cut it)
    bool whole;       // the tile holds the whole line
};
struct LineOut {
    bool tracked;     // the line took part in the walk (it has an info block in its header bytes: group_emit looks at it)
    uint32_t slots;   // wave-uniform: pair slots of the tile
    bool handled;     // false: the caller walks the line byte-wise
    uint32_t status;  // E_OK or the reference's error
    uint32_t n_pairs, n_ent;
    uint32_t msg_at;  // line index of the ' ' that starts the message (status == E_OK)
};

FG_WV uint32_t items_before(const Lds& L, uint32_t pos) {
    const uint64_t* Q = reinterpret_cast<const uint64_t*>(L.bmQ);
    return (uint32_t)L.qcnt[pos >> 6] + wv::popc64(Q[pos >> 6] & below(pos & 63u));
}
// the three quotes around pair slot (i0, j): the closing quote before it (or s0 - 1), its opening and its closing quote
FG_WV void pair_quotes(const Lds& L, uint32_t i0, uint32_t j, uint32_t s0, uint32_t* prevc, uint32_t* open, uint32_t* close) {
    const uint32_t ii = i0 + 2u * j;
    const uint16_t* it = L.items + ii;
    const uint32_t a = it[-1], b = it[0], c = it[1];  // (items[-1] exists)
    *prevc = j ? a : s0 - 1u;
    *open = b;
    *close = c;
}
FG_WV uint32_t rec_index(uint32_t i0, uint32_t j) { return ((i0 + (i0 & 1u)) >> 1) + j; }

// slot -> line, ONCE per tile: every line marks its first pair slot with the tile offset of its info block, then the marks are
// spread to the right, 64 slots per row (ballot + one cross-lane read per row, the last owner carried from row to row).
FG_WV void build_owner(const Lds& L, uint32_t total, bool line_has, uint32_t cum, uint32_t np, uint32_t hb) {
    const uint32_t lane = wv::lane();
    for (uint32_t p0 = 0; p0 < total; p0 += wv::kLanes) L.owner[p0 + lane] = 0xFFFFu;
    wv::sync();
    if (line_has && np != 0u) L.owner[cum] = (uint16_t)hb;
    wv::sync();
    uint32_t carry = 0;
    for (uint32_t p0 = 0; p0 < total; p0 += wv::kLanes) {
        const uint32_t v = L.owner[p0 + lane];
        const uint64_t marks = wv::ballot(v != 0xFFFFu);
        const uint64_t upto = marks & (lane == 63u ? ~0ull : ((2ull << lane) - 1ull));
        const uint32_t src = upto ? 63u - wv::clz64(upto) : 0u;
        const uint32_t got = wv::shfl(v, src);
        const uint32_t val = upto ? got : carry;
        L.owner[p0 + lane] = (uint16_t)val;
        carry = wv::bcast(val, 63u);
    }
    wv::sync();
}

// ---------------------------------------------------------------------------------------------
// The group step.  Called by all 64 lanes in wave-uniform control flow, after classify_tile().
// ---------------------------------------------------------------------------------------------
// (PROF: measurement build -- cycles of  [2] word + line pass + slot table  [3] P1  [4] P2  [5] verdict  are added to pc[])
template <bool PROF = false>
FG_WV LineOut group_walk(const Lds& L, uint32_t span, const LineIn& in, uint64_t* pc = nullptr) {
    const uint32_t lane = wv::lane();
    uint64_t tk = PROF ? wv::clock() : 0;
    auto tick = [&](int k) {
        if (PROF) {
            const uint64_t now = wv::clock();
            pc[k] += now - tk;
            tk = now;
        }
    };
    LineOut out{false, 0u, false, E_OK, 0u, 0u, 0u};
    const uint64_t* Q = reinterpret_cast<const uint64_t*>(L.bmQ);

    // ================= word pass: quote counts per word, positions of all real quotes =================
    const uint32_t nwords = (span + 63u) >> 6;
    const uint32_t wn = (nwords + wv::kLanes - 1u) / wv::kLanes;
    const uint32_t w0 = lane * wn < nwords ? lane * wn : nwords;
    const uint32_t w1 = w0 + wn < nwords ? w0 + wn : nwords;
    uint32_t n_items = 0;
    {
        uint32_t cnt = 0;
        for (uint32_t w = w0; w < w1; ++w) cnt += wv::popc64(Q[w]);
        uint32_t idx = wv::excl_sum(cnt, &n_items);
        const bool fits = n_items <= L.item_cap;  // wave-uniform
        for (uint32_t w = w0; w < w1; ++w) {
            L.qcnt[w] = (uint16_t)idx;
            uint64_t m = Q[w];
            while (m) {
                const uint32_t bit = wv::ctz64(m);
                m &= m - 1ull;
                if (fits) L.items[idx] = (uint16_t)(w * 64u + bit);
                ++idx;
            }
        }
        if (lane == 0u) {
            L.qcnt[nwords] = (uint16_t)n_items;
            L.qcnt[nwords + 1u] = (uint16_t)n_items;
            L.items[-1] = 0;
        }
    }
    wv::sync();
    const bool tile_ok = n_items <= L.item_cap && n_items <= 0xFFFFu;

    // ================= line pass =================
    const bool line_has = in.sd && tile_ok && in.d0 >= 32u && in.wlen > in.d0;  // (the info block needs 27 dead header bytes)
    const uint32_t s0 = in.base + in.d0, e = in.base + in.wlen;
    const uint32_t hb = (in.base + 3u) & ~3u;
    uint32_t i0 = 0, np = 0;
    if (line_has) {
        i0 = items_before(L, s0);
        const uint32_t i1 = items_before(L, e);
        np = (i1 - i0) >> 1;
        if (np > kMaxPairs) np = kMaxPairs;
    }
    uint32_t total = 0;
    const uint32_t cum = wv::excl_sum(np, &total);
    if (line_has) {
        uint32_t* info = L.tile_w + (hb >> 2);
        info[I_TV] = kNone;
        info[I_EL0] = 0u;
        info[I_EL1] = 0u;
        info[I_FIRST] = 0u;
        info[I_I0CUM] = i0 | (cum << 16);
        info[I_S0E] = s0 | (e << 16);
        info[I_BASEW] = in.base | ((in.whole ? 1u : 0u) << 16);
    }
    out.tracked = line_has;
    out.slots = total;
    build_owner(L, total, line_has, cum, np, hb);  // (ends with a barrier: the info blocks are visible too)

    tick(2);
    // ================= P1: every pair slot of the tile, 64 per trip =================
    uint32_t n_slow_all = 0;  // wave-uniform: the list's length lives in a register (no counter in LDS, no barrier per trip)
    for (uint32_t p0 = 0; p0 < total; p0 += wv::kLanes) {
        const bool act = p0 + lane < total;
        bool to_slow = false;
        if (act) {
            const uint32_t ohb = L.owner[p0 + lane];
            const uint32_t* oinfo = L.tile_w + (ohb >> 2);
            const uint32_t i0cum = oinfo[I_I0CUM], s0e = oinfo[I_S0E];
            const uint32_t oi0 = i0cum & 0xFFFFu, os0 = s0e & 0xFFFFu, oe = s0e >> 16;
            const uint32_t j = p0 + lane - (i0cum >> 16);
            uint32_t prevc, open, close;
            pair_quotes(L, oi0, j, os0, &prevc, &open, &close);
            const uint32_t gs = prevc + 1u, g = open - gs;
            // ---- the everyday pair: ` name=` in the 16 bytes before the opening quote
            uint32_t w[4];
            L.T.load16(open - 16u, w);  // (open >= s0 + 1 >= 33)
            const uint32_t nm = wv::gather16(name_flags(w[0]), name_flags(w[1]), name_flags(w[2]), name_flags(w[3]));
            // name = the run of name characters that ends at byte 14 (byte 15 must be the '=')
            const uint32_t not_nm = (~nm & 0x7FFFu) << 17;               // bit 31 <=> byte 14 is no name character
            const uint32_t nl = not_nm ? (uint32_t)wv::clz64((uint64_t)not_nm << 32) : 15u;
            const bool eq_ok = (w[3] >> 24) == '=';
            const uint32_t spb = 14u - nl;  // window byte in front of the name (valid for nl <= 14)
            const uint32_t spc = nl <= 14u ? (w[spb >> 2] >> (8u * (spb & 3u))) & 0xFFu : 0u;
            const bool ok_a = j != 0u && eq_ok && nl >= 1u && nl <= 13u && g == nl + 2u && spc == ' ';
            // ---- behind the closing quote
            const uint64_t tailb = L.T.load8(close);
            const uint32_t c1 = (uint32_t)(tailb >> 8) & 0xFFu, c2 = (uint32_t)(tailb >> 16) & 0xFFu;
            const bool rb = close + 1u < oe && c1 == ']';
            const bool term = rb && close + 2u < oe && c2 == ' ';
            const bool next_elem = rb && close + 2u < oe && c2 == '[';
            // ---- a backslash inside the value?
            const uint32_t vlen = close - open - 1u;
            bool esc = (wv::window64(L.bmB, open + 1u) & below(vlen)) != 0ull;
            if (vlen > 64u) esc = esc || wv::any_bit(L.bmB, open + 65u, close);
            const uint32_t r = rec_index(oi0, j);
            if (ok_a && (!rb || term || next_elem)) {
                L.rec[r] = (uint16_t)(nl | ((esc ? 1u : 0u) << 6) | (K_PAIR << 8));
                if (term) wv::lds_min(&L.tile_w[(ohb >> 2) + I_TV], 2u * (j + 1u) + 1u);
            } else {
                L.rec[r] = (uint16_t)(((esc ? 1u : 0u) << 6) | (K_BAIL << 8));
                to_slow = true;
            }
        }
        // the slots that need the exact walk, appended to the list
        const uint64_t sm = wv::ballot(to_slow);
        if (sm) {
            const uint32_t at = n_slow_all + wv::mbcnt(sm);
            if (to_slow && at < L.slow_cap) L.slow[at] = (uint16_t)(p0 + lane);  // (a list that overflows gives the whole tile back)
            n_slow_all += wv::popc64(sm);
        }
    }
    wv::sync();

    tick(3);
    // ================= P2: the exact state machine over the listed slots =================
    // (the shuffles are executed by all lanes: the per-lane work is written without early exits)
    {
        const bool list_overflow = n_slow_all > L.slow_cap;
        const uint32_t n_slow = list_overflow ? L.slow_cap : n_slow_all;
        for (uint32_t q0 = 0; q0 < n_slow; q0 += wv::kLanes) {
            const uint32_t q = q0 + lane;
            const bool act = q < n_slow;
            if (act) {
                const uint32_t p = L.slow[q];
                uint32_t* info = L.tile_w + ((uint32_t)L.owner[p] >> 2);
                const uint32_t i0cum = info[I_I0CUM], s0e = info[I_S0E];
                const bool kwhole = (info[I_BASEW] >> 16) & 1u;
                const uint32_t oi0 = i0cum & 0xFFFFu, ocum = i0cum >> 16, os0 = s0e & 0xFFFFu, oe = s0e >> 16;
                const uint32_t j = p - ocum;
                uint32_t prevc, open, close;
                pair_quotes(L, oi0, j, os0, &prevc, &open, &close);
                const uint32_t r = rec_index(oi0, j);
                const uint32_t esc = (L.rec[r] >> 6) & 1u;
                // (1) the bytes between the closing quote before and this opening quote: the everyday element start from masks, anything
                //     else by the exact walk
                Event ev{K_PAIR, E_OK, 0u, elem_start_shape(L.T, j ? prevc + 1u : os0, open, j == 0u), 1u};
                if (ev.name_len == 0u) ev = gap_walk(L.T, j ? prevc + 1u : os0, j == 0u, open, oe, kwhole);
                if (ev.kind == K_PAIR) {
                    L.rec[r] = (uint16_t)(ev.name_len | (esc << 6) | (ev.elem << 7) | (K_PAIR << 8));
                    if (ev.elem) wv::lds_or(&info[j < 32u ? I_EL0 : I_EL1], 1u << (j & 31u));
                    // (2) behind this pair's closing quote: `"]` + end of line / garbage are errors of THIS position
                    const uint64_t tailb = L.T.load8(close);
                    const uint32_t c1 = (uint32_t)(tailb >> 8) & 0xFFu, c2 = (uint32_t)(tailb >> 16) & 0xFFu;
                    if (close + 1u < oe && c1 == ']') {
                        if (close + 2u >= oe) {
                            if (kwhole) {
                                L.rec[r] |= (uint16_t)(E_NOMSG << 10);  // (payload of the event 2(j+1): read by the verdict)
                                wv::lds_min(&info[I_TV], 2u * (j + 1u));
                            } else {
                                wv::lds_min(&info[I_TV], 0u);  // the head ends here: give the line back (event 0 without an error payload)
                            }
                        } else if (c2 == ' ') {
                            wv::lds_min(&info[I_TV], 2u * (j + 1u) + 1u);
                        } else if (c2 != '[') {
                            L.rec[r] |= (uint16_t)(E_MALFORMED << 10);
                            wv::lds_min(&info[I_TV], 2u * (j + 1u));
                        }
                    }
                } else if (ev.kind == K_TERM) {
                    L.rec[r] = (uint16_t)((K_TERM << 8) | ((ev.msg_at - (j ? prevc + 1u : os0)) << 10));
                    wv::lds_min(&info[I_TV], 2u * j + 1u);
                } else if (ev.kind == K_ERR) {
                    L.rec[r] = (uint16_t)((K_ERR << 8) | (ev.status << 10));
                    wv::lds_min(&info[I_TV], 2u * j);
                } else {
                    L.rec[r] = (uint16_t)(K_BAIL << 8);
                    wv::lds_min(&info[I_TV], 2u * j);
                }
            }
        }
        wv::sync();
        tick(4);
        // ================= verdict (lane = line) =================
        if (line_has && !list_overflow) {
            uint32_t* info = L.tile_w + (hb >> 2);
            uint32_t tv = info[I_TV];
            const uint64_t elmask = (uint64_t)info[I_EL0] | ((uint64_t)info[I_EL1] << 32);
            uint32_t J = 0, msg_pos = 0, status = E_OK;
            bool good = false, decided = false;
            if (tv == kNone) {
                // no event among the pair slots: the tail behind the last pair (or the whole of it when there is no pair)
                if (np < kMaxPairs) {
                    uint32_t from = s0;
                    if (np) {
                        uint32_t prevc, open, close;
                        pair_quotes(L, i0, np - 1u, s0, &prevc, &open, &close);
                        from = close + 1u;
                    }
                    // (an unpaired quote behind the last pair -- a value that never closes -- ends the walk as K_BAIL)
                    const Event ev = gap_walk(L.T, from, np == 0u, kNone, e, in.whole);
                    if (ev.kind == K_TERM) {
                        J = np;
                        msg_pos = ev.msg_at;
                        good = decided = np != 0u;  // (`[id ] msg`: an element without pairs is not fast-form material)
                    } else if (ev.kind == K_ERR) {
                        status = ev.status;
                        decided = true;
                    }
                }
            } else if (tv & 1u) {
                J = tv >> 1;  // the structured data ends behind J pairs
                decided = good = J >= 1u;
                if (good) {
                    uint32_t prevc, open, close;
                    pair_quotes(L, i0, J - 1u, s0, &prevc, &open, &close);
                    // terminal right behind pair J-1 (`"] `), or inside the stretch of slot J (spaces before the bracket)
                    const uint32_t rj = J < np ? (uint32_t)L.rec[rec_index(i0, J)] : 0u;
                    if (J < np && ((rj >> 8) & 3u) == K_TERM) msg_pos = close + 1u + (rj >> 10);
                    else msg_pos = close + 2u;
                }
            } else {
                const uint32_t jj = tv >> 1;  // the first event is an error (or something only the byte-wise walker knows)
                // event 2j comes either from slot j's stretch (K_ERR / K_BAIL in its record) or from behind pair j-1's closing quote
                // (the error payload sits in pair j-1's record, whose kind is K_PAIR)
                const uint32_t rj = jj < np ? (uint32_t)L.rec[rec_index(i0, jj)] : 0u;
                const uint32_t rprev = jj >= 1u ? (uint32_t)L.rec[rec_index(i0, jj - 1u)] : 0u;
                if (jj >= 1u && ((rprev >> 8) & 3u) == K_PAIR && (rprev >> 10) != 0u) {
                    status = rprev >> 10;
                    decided = true;
                } else if (jj < np && ((rj >> 8) & 3u) == K_ERR) {
                    status = rj >> 10;
                    decided = true;
                }
            }
            // (every slot before the first event is a proven pair: an error, a bail or a terminal in an earlier stretch would have been
            //  the smaller event.)  The first pair opens the first element:
            if (good) decided = good = (elmask & 1ull) != 0ull;
            if (decided) {
                out.handled = true;
                out.status = status;
                if (good) {
                    out.n_pairs = J;
                    out.n_ent = J + wv::popc64(elmask & below(J));
                    out.msg_at = msg_pos - in.base;
                    info[I_TV] = J;  // (reused: the pairs to emit)
                } else {
                    info[I_TV] = 0u;
                }
            } else {
                info[I_TV] = 0u;
            }
        } else if (line_has) {
            L.tile_w[(hb >> 2) + I_TV] = 0u;
        }
    }
    wv::sync();
    tick(5);
    return out;
}

// ---------------------------------------------------------------------------------------------
// emit: the entries of the handled lines, one pair slot per lane.  `first` = the line's first entry slot (lane = line);
// emit_line = this lane's line was handled with status E_OK and its entries got slots.
// ---------------------------------------------------------------------------------------------
FG_WV void group_emit(const Lds& L, const DevTables& t, const LineIn& in, const LineOut& lo, bool emit_line, uint32_t first) {
    const uint32_t lane = wv::lane();
    if (lo.tracked) {  // this lane's line has an info block: the pairs to emit (none unless it was handled, is Ok and got its slots)
        uint32_t* info = L.tile_w + (((in.base + 3u) & ~3u) >> 2);
        info[I_FIRST] = first;
        if (!(lo.handled && lo.status == E_OK && emit_line)) info[I_TV] = 0u;
    }
    wv::sync();
    const uint32_t total = lo.slots;  // wave-uniform: the pair slots of the tile, as P1 walked them
    for (uint32_t p0 = 0; p0 < total; p0 += wv::kLanes) {
        if (p0 + lane >= total) continue;
        const uint32_t ohb = L.owner[p0 + lane];
        const uint32_t* info = L.tile_w + (ohb >> 2);
        const uint32_t i0cum = info[I_I0CUM];
        const uint32_t j = p0 + lane - (i0cum >> 16);
        if (j >= info[I_TV]) continue;  // (I_TV: the line's pairs, 0 for a line that is not emitted)
        const uint32_t oi0 = i0cum & 0xFFFFu, os0 = info[I_S0E] & 0xFFFFu, obase = info[I_BASEW] & 0xFFFFu;
        const uint64_t elmask = (uint64_t)info[I_EL0] | ((uint64_t)info[I_EL1] << 32);
        const uint32_t slot = info[I_FIRST] + j + wv::popc64(elmask & below(j + 1u));
        uint32_t prevc, open, close;
        pair_quotes(L, oi0, j, os0, &prevc, &open, &close);
        const uint32_t r = L.rec[rec_index(oi0, j)];
        const uint32_t nl = r & 63u, esc = (r >> 6) & 1u, elem = (r >> 7) & 1u;
        const uint32_t name_s = open - 1u - nl;
        gstore(t.ent_name, slot, fg_span{name_s - obase, nl});
        gstore(t.ent_val, slot, (uint64_t)(open + 1u - obase) | ((uint64_t)(close - open - 1u) << 32));
        gstore(t.ent_type, slot, (uint8_t)FG_T_STRING);
        gstore(t.ent_flags, slot, (uint8_t)(esc ? FG_EF_VAL_ESC : 0));
        if (elem) {  // the element this pair opens: `[id name=` (the line's first) or `][id name=`
            const uint32_t id_s = j ? prevc + 3u : os0 + 1u;
            gstore(t.ent_name, slot - 1u, fg_span{id_s - obase, name_s - 1u - id_s});
            gstore(t.ent_val, slot - 1u, 0ull);
            gstore(t.ent_type, slot - 1u, (uint8_t)FG_T_SDID);
            gstore(t.ent_flags, slot - 1u, (uint8_t)0);
        }
    }
}

}  // namespace sd2
}  // namespace fg
