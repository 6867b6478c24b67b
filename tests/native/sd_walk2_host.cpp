// Host build of flowgger_amd/csrc/fg_sd_walk2.hpp (the staged next form of the RFC5424 kernel's structured-data walk): stages
// groups of lines into a tile the way the streaming pipeline does (consecutive lines from a 16-byte boundary, quote/backslash
// bitmap per 16-byte chunk, garbage behind the staged bytes), runs the walk lane by lane and hands back what the kernel would
// derive from it.  Test infrastructure only.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../flowgger_amd/csrc/fg_sd_walk2.hpp"

namespace {
using namespace fg;

// the reference's state machine, byte by byte (rfc5424_decoder.rs:134-158, :174-242) -- the independent check of the walk
struct Ent { uint32_t name_s, name_len, val_s, val_len, esc, sdid; };
uint32_t byte_walk(const uint8_t* ln, uint32_t pos, uint32_t len, uint32_t* msg_at, std::vector<Ent>& out) {
    for (;;) {
        uint32_t s = pos + 1, sp = s;
        while (sp < len && ln[sp] != ' ') ++sp;
        if (sp >= len) return sd2::E_NOSD;
        out.push_back({s, sp - s, 0, 0, 0, 1});
        uint32_t st = 0, name_s = 0, name_e = 0, val_s = 0, esc = 0, after = 0;
        for (uint32_t i = sp + 1; i < len; ++i) {
            const uint32_t c = ln[i];
            if (st == 3) {
                if (c == '\\') { st = 4; esc = 1; }
                else if (c == '"') { out.push_back({name_s, name_e - name_s, val_s, i - val_s, esc, 0}); st = 0; }
            } else if (st == 4) {
                st = 3;
            } else {
                const bool is_name = sd2::is_name_char(c);
                if (st == 0) {
                    if (c == ' ' || c == '"') {}
                    else if (c == ']') { after = i + 1; break; }
                    else if (is_name) { st = 1; name_s = i; }
                    else return sd2::E_SDFMT;
                } else if (st == 1) {
                    if (is_name) {}
                    else if (c == '=') { name_e = i; st = 2; }
                    else return sd2::E_SDFMT;
                } else {
                    if (c != '"') return sd2::E_SDFMT;
                    st = 3; val_s = i + 1; esc = 0;
                }
            }
        }
        if (after == 0) return sd2::E_NOBRACKET;
        if (after >= len) return sd2::E_NOMSG;
        if (ln[after] == '[') { pos = after; continue; }
        if (ln[after] != ' ') return sd2::E_MALFORMED;
        *msg_at = after;
        return sd2::E_OK;
    }
}
}  // namespace

// For every line i (bytes[offsets[i], offsets[i+1]), structured data starting at line index sd_pos[i]):
//   status[i], msg_at[i], n_ent[i], rec_ok[i] from the walk in SD_STASH mode (lean = 1: the lean step on; 0: general steps only);
//   its entries (from the records left in the tile, or -- rec_ok == 0 -- from an SD_EMIT walk over a fresh copy) as six uint32
//   per entry in ent[6 * (ent_first[i] + k)] = name_s, name_len, val_s, val_len, esc, sdid;
//   ref_status[i] / ref_* the same from the byte-wise state machine.  Returns the number of entries, or -1 when ent_cap is too small.
// The SD_COUNT walk is run as well and must agree (return -2 - i when it does not).
extern "C" void fgs_stats(unsigned long long out[4], int reset) {
    for (int k = 0; k < fg::sd2::ST_N; ++k) {
        out[k] = fg::sd2::stats()[k];
        if (reset) fg::sd2::stats()[k] = 0;
    }
}

extern "C" long fgs_walk(const uint8_t* bytes, uint64_t nbytes, const uint64_t* offsets, uint64_t n, const uint32_t* sd_pos, uint32_t lines_per_group,
                         int lean, uint32_t* status, uint32_t* msg_at, uint32_t* n_ent, uint8_t* rec_ok, uint32_t* ent_first, uint32_t* ent,
                         uint64_t ent_cap, uint32_t* ref_status, uint32_t* ref_msg_at, uint32_t* ref_ent_first, uint32_t* ref_n_ent, uint32_t* ref_ent) {
    uint64_t used = 0, ref_used = 0;
    const uint32_t L = lines_per_group;
    for (uint64_t g0 = 0; g0 < n; g0 += L) {
        const uint32_t nl = (uint32_t)(g0 + L <= n ? L : n - g0);
        const uint64_t a0 = offsets[g0] & ~15ull;
        const uint32_t span = (uint32_t)((offsets[g0 + nl] - a0 + 15ull) & ~15ull);
        // tile: span bytes + 64 of padding (garbage on purpose), bitmap: one bit per byte + three dwords of padding
        std::vector<uint32_t> tile_words(span / 4 + 16), bm_words(span / 32 + 4);
        auto stage = [&]() {
            uint8_t* tb = reinterpret_cast<uint8_t*>(tile_words.data());
            memset(tb, 0xA5, tile_words.size() * 4);
            for (uint32_t k = 0; k < span; ++k) tb[k] = a0 + k < nbytes ? bytes[a0 + k] : 0x22;  // (quotes behind the batch: must not matter)
            memset(bm_words.data(), 0xFF, bm_words.size() * 4);                                // (set bits behind the tile: must not matter)
            uint16_t* bm16 = reinterpret_cast<uint16_t*>(bm_words.data());
            for (uint32_t c = 0; c < span / 16; ++c) {
                uint32_t m = 0;
                for (uint32_t k = 0; k < 16; ++k) m |= (tb[16 * c + k] == '"' || tb[16 * c + k] == '\\') ? 1u << k : 0u;
                bm16[c] = (uint16_t)m;
            }
        };
        stage();
        for (uint32_t j = 0; j < nl; ++j) {
            const uint64_t i = g0 + j;
            const uint32_t base = (uint32_t)(offsets[i] - a0), len = (uint32_t)(offsets[i + 1] - offsets[i]);
            sd2::Tile T{wv::Bytes{tile_words.data()}, bm_words.data()};
            DevTables none{};
            // ---- reference
            std::vector<Ent> re;
            uint32_t rm = 0;
            ref_status[i] = byte_walk(bytes + offsets[i], sd_pos[i], len, &rm, re);
            ref_msg_at[i] = ref_status[i] == sd2::E_OK ? rm : 0;
            ref_ent_first[i] = (uint32_t)ref_used;
            ref_n_ent[i] = ref_status[i] == sd2::E_OK ? (uint32_t)re.size() : 0;
            if (ref_used + ref_n_ent[i] > ent_cap) return -1;
            for (uint32_t k = 0; k < ref_n_ent[i]; ++k) {
                const uint32_t v[6] = {re[k].name_s, re[k].name_len, re[k].val_s, re[k].val_len, re[k].esc, re[k].sdid};
                memcpy(ref_ent + 6 * (ref_used + k), v, sizeof v);
            }
            ref_used += ref_n_ent[i];
            // ---- count mode (the tile is intact)
            uint32_t cm = 0, cn = 0;
            const uint32_t cst = lean ? sd2::walk<sd2::SD_COUNT, true>(T, base, sd_pos[i], len, &cm, &cn, none, 0)
                                      : sd2::walk<sd2::SD_COUNT, false>(T, base, sd_pos[i], len, &cm, &cn, none, 0);
            // ---- stash mode
            uint32_t m = 0, ne = 0;
            bool ok = false;
            uint32_t* tw = tile_words.data();
            const uint32_t st = lean ? sd2::walk<sd2::SD_STASH, true>(T, base, sd_pos[i], len, &m, &ne, none, 0, tw, &ok)
                                     : sd2::walk<sd2::SD_STASH, false>(T, base, sd_pos[i], len, &m, &ne, none, 0, tw, &ok);
            if (cst != st || (st == sd2::E_OK && (cm != m || cn != ne))) return -2 - (long)i;
            status[i] = st;
            msg_at[i] = st == sd2::E_OK ? m : 0;
            n_ent[i] = st == sd2::E_OK ? ne : 0;
            rec_ok[i] = st == sd2::E_OK && ok;
            ent_first[i] = (uint32_t)used;
            if (used + n_ent[i] > ent_cap) return -1;
            if (st == sd2::E_OK && ok) {
                const uint32_t* rec32 = tw + (((base + 3u) & ~3u) >> 2);
                for (uint32_t k = 0; k < ne; ++k) {
                    const uint32_t lo = rec32[2 * k], hi = rec32[2 * k + 1];
                    const uint32_t name_s = lo & 0xFFFFu, name_len = lo >> 16, val_len = hi & 0xFFFFu, sdid = (hi >> 17) & 1u;
                    const uint32_t v[6] = {name_s, name_len, sdid ? 0u : name_s + name_len + 2u, sdid ? 0u : val_len, (hi >> 16) & 1u, sdid};
                    memcpy(ent + 6 * (used + k), v, sizeof v);
                }
            } else if (st == sd2::E_OK) {
                // the line's copy in the tile is no longer intact: stage again, emit mode into plain arrays
                stage();
                std::vector<fg_span> en(ne);
                std::vector<uint64_t> ev(ne);
                std::vector<uint8_t> ety(ne), efl(ne);
                DevTables t{};
                t.ent_name = en.data(); t.ent_val = ev.data(); t.ent_type = ety.data(); t.ent_flags = efl.data();
                uint32_t m2 = 0, ne2 = 0;
                const uint32_t st2 = lean ? sd2::walk<sd2::SD_EMIT, true>(T, base, sd_pos[i], len, &m2, &ne2, t, 0)
                                          : sd2::walk<sd2::SD_EMIT, false>(T, base, sd_pos[i], len, &m2, &ne2, t, 0);
                if (st2 != st || m2 != m || ne2 != ne) return -2 - (long)i;
                for (uint32_t k = 0; k < ne; ++k) {
                    const uint32_t sdid = ety[k] == FG_T_SDID;
                    const uint32_t v[6] = {en[k].off, en[k].len, sdid ? 0u : (uint32_t)ev[k], sdid ? 0u : (uint32_t)(ev[k] >> 32),
                                           (uint32_t)(efl[k] & FG_EF_VAL_ESC ? 1 : 0), sdid};
                    memcpy(ent + 6 * (used + k), v, sizeof v);
                }
            }
            used += n_ent[i];
        }
    }
    return (long)used;
}

// Two lanes per line, the adversarial order: the SECOND lane runs first (from the guessed split, leaving its records in the tile),
// then the first lane with stop_at.  out[i]: 0 = no split offered, 1 = hand-over (combined result written to status / msg_at / n_ent
// / ent as in fgs_walk), 2 = E_REDO (the guess did not hold; nothing written), 3 = the first lane finished on its own (an error
// before the split: its result is written).  split[i] = the guess.  Returns the number of entries or -1.
extern "C" long fgs_walk_two(const uint8_t* bytes, uint64_t nbytes, const uint64_t* offsets, uint64_t n, const uint32_t* sd_pos,
                             uint32_t lines_per_group, uint32_t* status, uint32_t* msg_at, uint32_t* n_ent, uint32_t* ent_first, uint32_t* ent,
                             uint64_t ent_cap, uint8_t* out, uint32_t* split) {
    uint64_t used = 0;
    const uint32_t L = lines_per_group;
    for (uint64_t g0 = 0; g0 < n; g0 += L) {
        const uint32_t nl = (uint32_t)(g0 + L <= n ? L : n - g0);
        const uint64_t a0 = offsets[g0] & ~15ull;
        const uint32_t span = (uint32_t)((offsets[g0 + nl] - a0 + 15ull) & ~15ull);
        std::vector<uint32_t> tile_words(span / 4 + 16), bm_words(span / 32 + 20);
        uint8_t* tb = reinterpret_cast<uint8_t*>(tile_words.data());
        memset(tb, 0xA5, tile_words.size() * 4);
        for (uint32_t k = 0; k < span; ++k) tb[k] = a0 + k < nbytes ? bytes[a0 + k] : 0x22;
        memset(bm_words.data(), 0xFF, bm_words.size() * 4);
        uint16_t* bm16 = reinterpret_cast<uint16_t*>(bm_words.data());
        for (uint32_t c = 0; c < span / 16; ++c) {
            uint32_t m = 0;
            for (uint32_t k = 0; k < 16; ++k) m |= (tb[16 * c + k] == '"' || tb[16 * c + k] == '\\') ? 1u << k : 0u;
            bm16[c] = (uint16_t)m;
        }
        uint32_t* tw = tile_words.data();
        sd2::Tile T{wv::Bytes{tile_words.data()}, bm_words.data()};
        DevTables none{};
        for (uint32_t j = 0; j < nl; ++j) {
            const uint64_t i = g0 + j;
            const uint32_t base = (uint32_t)(offsets[i] - a0), len = (uint32_t)(offsets[i + 1] - offsets[i]);
            status[i] = msg_at[i] = n_ent[i] = 0;
            ent_first[i] = (uint32_t)used;
            const uint32_t sp = sd2::pick_split(T, base, sd_pos[i], len);
            split[i] = sp;
            out[i] = 0;
            if (sp == sd2::kNoSplit || sp + 1u >= len) continue;
            // second lane first
            uint32_t mB = 0, nB = 0;
            bool okB = false;
            const uint32_t stB = sd2::walk<sd2::SD_STASH, true>(T, base, 0u, len, &mB, &nB, none, 0, tw, &okB, sd2::kNoSplit, sp + 1u);
            // first lane
            uint32_t mA = 0, nA = 0;
            bool okA = false;
            const uint32_t stA = sd2::walk<sd2::SD_STASH, true>(T, base, sd_pos[i], len, &mA, &nA, none, 0, tw, &okA, sp);
            if (stA == sd2::E_REDO) {
                out[i] = 2;
                continue;
            }
            if (stA != sd2::E_HANDOFF) {  // an outcome of its own (must be an error: the OK end lies behind the split)
                out[i] = 3;
                status[i] = stA;
                msg_at[i] = stA == sd2::E_OK ? mA : 0;
                continue;
            }
            out[i] = 1;
            status[i] = stB;
            if (stB != sd2::E_OK) continue;
            msg_at[i] = mB;
            n_ent[i] = nA + nB;
            if (used + n_ent[i] > ent_cap) return -1;
            if (!(okA && okB)) {  // (records did not fit: the kernel walks the line again in global memory; nothing to compare here)
                out[i] = 4;
                n_ent[i] = 0;
                continue;
            }
            for (int half = 0; half < 2; ++half) {
                const uint32_t* rec32 = tw + (((base + (half ? sp + 1u : 0u) + 3u) & ~3u) >> 2);
                const uint32_t cnt = half ? nB : nA, at = half ? nA : 0u;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const uint32_t lo = rec32[2 * k], hi = rec32[2 * k + 1];
                    const uint32_t name_s = lo & 0xFFFFu, name_len = lo >> 16, val_len = hi & 0xFFFFu, sdid = (hi >> 17) & 1u;
                    const uint32_t v[6] = {name_s, name_len, sdid ? 0u : name_s + name_len + 2u, sdid ? 0u : val_len, (hi >> 16) & 1u, sdid};
                    memcpy(ent + 6 * (used + at + k), v, sizeof v);
                }
            }
            used += n_ent[i];
        }
    }
    return (long)used;
}


// The whole group through sd2::walk_group + wave prefix sum + sd2::copy_out on the fiber emulation of a wavefront (what stage B of
// the kernel will do): fills status / msg_at / n_ent / ent_first and the entry arrays (ent_name, ent_val, ent_type, ent_flags of
// `tables`); lines the group walk hands back (redo, records that did not fit) are walked alone in EMIT mode over a fresh copy, as
// the kernel does from global memory.  kinds[i]: 0 = records copied out by one lane, 1 = by two lanes, 2 = redo, 3 = records did
// not fit, 4 = not OK.  Returns entries used or -1.
extern "C" long fgs_walk_wave(const uint8_t* bytes, uint64_t nbytes, const uint64_t* offsets, uint64_t n, const uint32_t* sd_pos,
                              uint32_t lines_per_group, const fg_tables* tables, uint32_t* status, uint32_t* msg_at, uint8_t* kinds) {
    using namespace fg;
    DevTables t{};
    t.ent_cap = tables->ent_cap;
    t.ent_name = tables->ent_name;
    t.ent_val = tables->ent_val;
    t.ent_type = tables->ent_type;
    t.ent_flags = tables->ent_flags;
    uint32_t used = 0;
    const uint32_t L = lines_per_group;
    for (uint64_t g0 = 0; g0 < n; g0 += L) {
        const uint32_t nl = (uint32_t)(g0 + L <= n ? L : n - g0);
        const uint64_t a0 = offsets[g0] & ~15ull;
        const uint32_t span = (uint32_t)((offsets[g0 + nl] - a0 + 15ull) & ~15ull);
        std::vector<uint32_t> tile_words(span / 4 + 16), bm_words(span / 32 + 4);
        auto stage = [&]() {
            uint8_t* tb = reinterpret_cast<uint8_t*>(tile_words.data());
            memset(tb, 0xA5, tile_words.size() * 4);
            for (uint32_t k = 0; k < span; ++k) tb[k] = a0 + k < nbytes ? bytes[a0 + k] : 0x22;
            memset(bm_words.data(), 0xFF, bm_words.size() * 4);
            uint16_t* bm16 = reinterpret_cast<uint16_t*>(bm_words.data());
            for (uint32_t c = 0; c < span / 16; ++c) {
                uint32_t m = 0;
                for (uint32_t k = 0; k < 16; ++k) m |= (tb[16 * c + k] == '"' || tb[16 * c + k] == '\\') ? 1u << k : 0u;
                bm16[c] = (uint16_t)m;
            }
        };
        stage();
        sd2::Tile T{wv::Bytes{tile_words.data()}, bm_words.data()};
        uint32_t* tw = tile_words.data();
        sd2::PairOut outs[64];
        uint32_t firsts[64];
        uint32_t group_total = 0;
        const bool two = nl <= 32;
        emu::run_wave([&]() {
            const uint32_t l = wv::lane();
            const bool has = l < nl;
            const uint32_t base = has ? (uint32_t)(offsets[g0 + l] - a0) : 0u, len = has ? (uint32_t)(offsets[g0 + l + 1] - offsets[g0 + l]) : 0u;
            const sd2::PairOut o = sd2::walk_group<sd2::SD_STASH>(T, tw, two, has, base, has ? sd_pos[g0 + l] : 0u, len, t);
            // slots: the line's lane asks for the line's entries (only lines whose records can be copied out take part here)
            const uint32_t want = (has && !o.redo && o.status == sd2::E_OK && o.rec_ok) ? o.n_ent : 0u;
            uint32_t total = 0;
            const uint32_t ex = wv::excl_sum(want, &total);
            const uint32_t first = used + ex;
            const uint32_t line_first = wv::shfl(first, l & 31u);
            sd2::copy_out(tw, o.rec_at, o.n_own, (two ? line_first : first) + o.skip, t);
            outs[l] = o;
            firsts[l] = first;
            if (l == 0) group_total = total;
        });
        used += group_total;
        for (uint32_t j = 0; j < nl; ++j) {
            const uint64_t i = g0 + j;
            const sd2::PairOut& o = outs[j];
            const uint32_t base = (uint32_t)(offsets[i] - a0), len = (uint32_t)(offsets[i + 1] - offsets[i]);
            if (!o.redo && o.status == sd2::E_OK && o.rec_ok) {
                status[i] = sd2::E_OK;
                msg_at[i] = o.msg_at;
                tables->ent_first[i] = firsts[j];
                tables->ent_count[i] = o.n_ent;
                kinds[i] = (two && outs[j + 32].n_own) ? 1 : 0;
                continue;
            }
            if (!o.redo && o.status != sd2::E_OK) {
                status[i] = o.status;
                msg_at[i] = 0;
                tables->ent_first[i] = 0;
                tables->ent_count[i] = 0;
                kinds[i] = 4;
                continue;
            }
            // handed back: alone, over a fresh copy, entries straight into the table
            kinds[i] = o.redo ? 2 : 3;
            stage();
            uint32_t m = 0, cnt = 0;
            status[i] = sd2::walk<sd2::SD_COUNT, true>(T, base, sd_pos[i], len, &m, &cnt, t, 0);
            msg_at[i] = status[i] == sd2::E_OK ? m : 0;
            if (status[i] != sd2::E_OK) cnt = 0;
            if (used + cnt > tables->ent_cap) return -1;
            if (cnt) sd2::walk<sd2::SD_EMIT, true>(T, base, sd_pos[i], len, &m, &cnt, t, used);
            tables->ent_first[i] = used;
            tables->ent_count[i] = cnt;
            used += cnt;
        }
    }
    *tables->ent_used = used;
    return (long)used;
}
