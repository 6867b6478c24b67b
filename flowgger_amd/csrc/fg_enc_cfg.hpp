// fg_enc_cfg.hpp -- host-side construction of the encoder kernels' configuration (EncCfg + its blob) from the
// C ABI's fg_encode_cfg: what XEncoder::new(&Config) prepares once (encoder/gelf_encoder.rs:16-38,
// ltsv_encoder.rs:11-30, mod.rs:58-79).  Header-only so that the CPU tests build the very same configuration.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "fg_emit.hpp"

namespace fg {

inline void align4(std::vector<uint8_t>* b) {
    while (b->size() & 3u) b->push_back(0);
}
struct EncCfgHost {
    std::vector<uint8_t> blob;
    std::vector<StaticKey> keys;
    EncCfg cfg{};  // blob / keys pointers are left null: the caller points them at its copy
};

// suffix[k] / has_suffix[k]: the LTSV decoder's type suffixes (bool, f64, i64, u64).  Returns false on bad arguments.
inline bool build_enc_cfg(fg_format src_fmt, const fg_encode_cfg* ec, const std::string suffix[4], const bool has_suffix[4],
                          EncCfgHost* out) {
    if (!ec || (int)ec->encoder < 0 || (int)ec->encoder > (int)FG_ENC_PASSTHROUGH) return false;
    if ((int)ec->merger < 0 || (int)ec->merger > (int)FG_MERGE_SYSLEN) return false;
    if (ec->n_extra && (!ec->extra_keys || !ec->extra_values)) return false;
    for (uint32_t i = 0; i < ec->n_extra; ++i)
        if (!ec->extra_keys[i] || !ec->extra_values[i]) return false;
    std::vector<uint8_t>& blob = out->blob;
    EncCfg& cfg = out->cfg;
    blob.clear();
    out->keys.clear();
    cfg = EncCfg{};
    if (ec->encoder == FG_ENC_GELF) {
        // the static part of every object: nine fixed keys, replaced / extended by output.gelf_extra (inserted last,
        // gelf_encoder.rs:107-109), in BTreeMap (byte) order
        struct Ent {
            uint32_t kind;
            std::string val;
        };
        std::map<std::string, Ent> m;
        const char* fixed[9] = {"application_name", "full_message", "host", "level", "process_id", "sd_id", "short_message", "timestamp", "version"};
        for (uint32_t k = 0; k < 9; ++k) m[fixed[k]] = Ent{k, ""};
        for (uint32_t i = 0; i < ec->n_extra; ++i) m[ec->extra_keys[i]] = Ent{(uint32_t)SK_EXTRA, ec->extra_values[i]};
        auto json_text = [](const std::string& x, std::string* o) {  // serde_json 0.8 escape_str
            static const char hex[] = "0123456789abcdef";
            o->push_back('"');
            for (unsigned char c : x) {
                switch (c) {
                    case '"': *o += "\\\""; break;
                    case '\\': *o += "\\\\"; break;
                    case 8: *o += "\\b"; break;
                    case 9: *o += "\\t"; break;
                    case 10: *o += "\\n"; break;
                    case 12: *o += "\\f"; break;
                    case 13: *o += "\\r"; break;
                    default:
                        if (c < 0x20) {
                            *o += "\\u00";
                            o->push_back(hex[c >> 4]);
                            o->push_back(hex[c & 15]);
                        } else {
                            o->push_back((char)c);
                        }
                }
            }
            o->push_back('"');
        };
        for (const auto& kv : m) {
            StaticKey k{};
            k.key_off = (uint32_t)blob.size();
            k.key_len = (uint32_t)kv.first.size();
            blob.insert(blob.end(), kv.first.begin(), kv.first.end());
            k.kind = kv.second.kind;
            // (StaticKey: the member's text with everything that is constant, behind ',' and behind '{')
            std::string text = ",";
            json_text(kv.first, &text);
            text.push_back(':');
            if (kv.second.kind == SK_EXTRA) json_text(kv.second.val, &text);
            else if (kv.second.kind == SK_VERSION) text += "\"1.1\"";
            else if (kv.second.kind != SK_LEVEL && kv.second.kind != SK_TS) text.push_back('"');  // a string: its opening quote
            k.text_len = (uint32_t)text.size();
            align4(&blob);
            k.text_off = (uint32_t)blob.size();
            blob.insert(blob.end(), text.begin(), text.end());
            text[0] = '{';
            align4(&blob);
            k.text1_off = (uint32_t)blob.size();
            blob.insert(blob.end(), text.begin(), text.end());
            out->keys.push_back(k);
        }
    }
    for (int k = 0; k < 4; ++k) {
        cfg.set_suffix((uint32_t)k, (uint32_t)blob.size(), has_suffix[k] ? (uint32_t)suffix[k].size() : 0xFFFFFFFFu);
        if (has_suffix[k]) blob.insert(blob.end(), suffix[k].begin(), suffix[k].end());
    }
    align4(&blob);
    cfg.ltsv_extra_off = (uint32_t)blob.size();
    if (ec->encoder == FG_ENC_LTSV) {
        // output.ltsv_extra through LTSVString::insert (ltsv_encoder.rs:37-58,96-103): one leading '_' stripped from
        // the key, key / value characters replaced, pairs joined with TAB
        for (uint32_t i = 0; i < ec->n_extra; ++i) {
            if (i) blob.push_back('\t');
            const char* k = ec->extra_keys[i];
            if (*k == '_') ++k;
            for (; *k; ++k) blob.push_back((uint8_t)(*k == '\n' || *k == '\t' ? ' ' : *k == ':' ? '_' : *k));
            blob.push_back(':');
            for (const char* v = ec->extra_values[i]; *v; ++v) blob.push_back((uint8_t)(*v == '\n' || *v == '\t' ? ' ' : *v));
        }
    }
    cfg.ltsv_extra_len = (uint32_t)blob.size() - cfg.ltsv_extra_off;
    align4(&blob);
    cfg.prepend_off = (uint32_t)blob.size();
    cfg.prepend_len = 0xFFFFFFFFu;
    if (ec->prepend && (ec->encoder == FG_ENC_RFC3164 || ec->encoder == FG_ENC_PASSTHROUGH)) {
        const std::string p = ec->prepend;
        blob.insert(blob.end(), p.begin(), p.end());
        cfg.prepend_len = (uint32_t)p.size();
    }
    blob.resize(blob.size() + 16, 0);  // readable padding behind the last piece (Base::blob reads sixteen bytes at a piece's tail)
    cfg.src_fmt = (uint32_t)src_fmt;
    cfg.enc = (uint32_t)ec->encoder;
    cfg.merger = (uint32_t)ec->merger;
    cfg.n_keys = (uint32_t)out->keys.size();
    cfg.now_ts = ec->now_ts;
    cfg.sort_slots = emit::kSortSlots;
    return true;
}

}  // namespace fg
