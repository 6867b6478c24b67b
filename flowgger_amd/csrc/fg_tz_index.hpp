// fg_tz_index.hpp -- host-side builder of the zone table the RFC3164 kernel searches (fg::r3164::TzView).
//
// Input: the fg_tz_table of the configuration (include/fg_hip.h: names sorted bytewise, per zone the spans
// (utc_start, utc_offset)), i.e. what time_tz::timezones::get_by_name + assume_timezone consult in the reference
// (rfc3164_decoder.rs:195-207).  Output: one blob (relocatable: all pointers of the view are blob offsets until
// view() binds them to a base address -- host memory for the CPU tests, HBM for the kernel) with
//   * the zone records and the names (dword aligned, zero padded) for a word-wise comparison,
//   * an open-addressing hash table name -> zone (load factor <= 1/2, 16-bit tag + zone + 1 per slot),
//   * register-resident reject masks (first byte, length) so that a hostname token rarely reaches the table,
//   * per zone the span range that covers the configured year, so that the offset search for a date without a year
//     (always the configured year) takes one or two probes instead of log2(spans).
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "fg_rfc3164_parse.hpp"

namespace fg {
namespace r3164 {

struct TzIndex {
    std::vector<uint8_t> blob;
    uint64_t o_zones = 0, o_names = 0, o_slots = 0, o_start = 0, o_off = 0;
    TzView proto{};  // masks, counts, hint window (pointers unset)

    // false: the table cannot be indexed (>= 65535 zones)
    bool build(const std::vector<std::string>& names, const std::vector<uint32_t>& zone_first, const std::vector<int64_t>& utc_start,
               const std::vector<int32_t>& utc_off, int32_t current_year) {
        const uint32_t nz = (uint32_t)names.size();
        *this = TzIndex{};
        if (nz == 0) return true;
        if (nz >= 65535u) return false;
        auto up = [](uint64_t v, uint64_t a) { return (v + a - 1) / a * a; };
        const int64_t hint_lo = days_from_civil(current_year, 1, 1) * 86400ll;
        const int64_t hint_hi = days_from_civil(current_year + 1, 1, 1) * 86400ll;
        std::vector<TzZone> zones(nz);
        std::vector<uint8_t> name_bytes;
        uint64_t first_lo = 0, first_hi = 0, len_mask = 0;
        uint32_t slots_n = 64;
        while (slots_n < 2u * nz) slots_n <<= 1;
        std::vector<uint32_t> slots(slots_n, 0u);
        for (uint32_t z = 0; z < nz; ++z) {
            const std::string& nm = names[z];
            TzZone& r = zones[z];
            r.name_off = (uint32_t)name_bytes.size();
            r.name_len = (uint32_t)nm.size();
            name_bytes.insert(name_bytes.end(), nm.begin(), nm.end());
            name_bytes.resize(up(name_bytes.size() + 1, 4), 0);  // >= 1 zero byte: an empty name still owns a dword
            r.first = zone_first[z];
            r.last = zone_first[z + 1] - 1u;
            // the linear rule of assume_timezone (first span whose local end lies after the time) for the two ends of
            // the hint window; a binary search inside [y_lo, y_hi] then gives the same span as one over [first, last]
            // when the local span ends increase monotonically -- otherwise the zone is not hinted
            bool monotone = true;
            for (uint32_t i = r.first; i + 1u < r.last; ++i)
                if (utc_start[i + 1] + utc_off[i] > utc_start[i + 2] + utc_off[i + 1]) monotone = false;
            auto linear = [&](int64_t local) {
                for (uint32_t i = r.first; i < r.last; ++i)
                    if (local < utc_start[i + 1] + (int64_t)utc_off[i]) return i;
                return r.last;
            };
            r.y_lo = monotone ? linear(hint_lo) : r.first;
            r.y_hi = monotone ? linear(hint_hi - 1) : r.last;
            if (!nm.empty()) {
                const uint8_t c0 = (uint8_t)nm[0];
                if (c0 < 64) first_lo |= 1ull << c0;
                else if (c0 < 128) first_hi |= 1ull << (c0 - 64);
            }
            len_mask |= 1ull << (nm.size() < 63 ? nm.size() : 63);
            uint32_t h = kTzHashInit;
            for (unsigned char c : nm) h = tz_hash_step(h, c);
            uint32_t slot = h & (slots_n - 1u);
            while (slots[slot]) slot = (slot + 1u) & (slots_n - 1u);
            slots[slot] = (h >> 16) << 16 | (z + 1u);
        }
        o_zones = 0;
        o_names = up(o_zones + nz * sizeof(TzZone), 16);
        o_slots = up(o_names + name_bytes.size(), 16);
        o_start = up(o_slots + slots_n * 4ull, 16);
        o_off = up(o_start + utc_start.size() * 8ull, 16);
        blob.assign(up(o_off + utc_off.size() * 4ull, 16), 0);
        memcpy(blob.data() + o_zones, zones.data(), nz * sizeof(TzZone));
        memcpy(blob.data() + o_names, name_bytes.data(), name_bytes.size());
        memcpy(blob.data() + o_slots, slots.data(), slots_n * 4ull);
        memcpy(blob.data() + o_start, utc_start.data(), utc_start.size() * 8ull);
        memcpy(blob.data() + o_off, utc_off.data(), utc_off.size() * 4ull);
        proto.nz = nz;
        proto.slot_mask = slots_n - 1u;
        proto.hint_lo = hint_lo;
        proto.hint_hi = hint_hi;
        proto.first_lo = first_lo;
        proto.first_hi = first_hi;
        proto.len_mask = len_mask;
        return true;
    }
    // the view with its pointers bound to a copy of the blob at `base` (16-byte aligned)
    TzView view(const uint8_t* base) const {
        TzView v = proto;
        if (v.nz == 0) return v;
        v.zones = reinterpret_cast<const TzZone*>(base + o_zones);
        v.name_words = reinterpret_cast<const uint32_t*>(base + o_names);
        v.slots = reinterpret_cast<const uint32_t*>(base + o_slots);
        v.utc_start = reinterpret_cast<const int64_t*>(base + o_start);
        v.utc_off = reinterpret_cast<const int32_t*>(base + o_off);
        return v;
    }
};

}  // namespace r3164
}  // namespace fg
