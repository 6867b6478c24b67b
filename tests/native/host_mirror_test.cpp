// Exercises the C++ host mirror (flowgger_amd/host/fg_decoder.hpp): frames a file like the
// reference splitters, decodes in batches on the GPU, prints one hex canonical Record per Ok line
// on stdout and the reference's error lines on stderr.  usage: host_mirror_test <rfc5424|ltsv|gelf> <line|nul|syslen|gpu-line|gpu-nul|pipe-line|pipe-nul|pipe-syslen> <file> [batch]
#include <cstdio>
#include <fstream>
#include <iostream>

#include "../../flowgger_amd/host/fg_decoder.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    std::string fmt = argv[1], fr = argv[2];
    std::unique_ptr<fg::Decoder> dec;
    if (fmt == "rfc5424") dec.reset(new fg::RFC5424Decoder());
    else if (fmt == "gelf") dec.reset(new fg::GelfDecoder());
    else {
        fg::LtsvConfig c;
        c.schema = {{"counter", FG_T_U64}, {"score", FG_T_I64}, {"mean", FG_T_F64}, {"done", FG_T_BOOL}};
        c.suffix_u64 = "_u64";
        c.suffix_f64 = "_f64";
        dec.reset(new fg::LTSVDecoder(c));
    }
    auto clone = dec->clone_boxed();  // per-connection clone, like tcp_input.rs:39-47
    auto hex_sink = [](fg::Record&& r) {
        fg::DecodeResult d;
        d.record = std::move(r);
        std::string c = fg::to_canonical(d);
        for (unsigned char ch : c) printf("%02x", ch);
        printf("\n");
    };
    if (fr == "pipe-line" || fr == "pipe-nul" || fr == "pipe-syslen") {  // the whole handle_line on the GPU: encoded GELF stream on stdout
        fg::EncoderConfig ec;
        ec.encoder = FG_ENC_GELF;
        ec.merger = fr == "pipe-line" ? FG_MERGE_LINE : fr == "pipe-nul" ? FG_MERGE_NUL : FG_MERGE_SYSLEN;
        ec.extra = {{"_site", "dc1"}, {"a_first", "x\"y"}};
        ec.now_ts = 1438859724.638;
        fg::TranscodingSplitter tsp(fr == "pipe-line" ? fg::TranscodingSplitter::Line : fr == "pipe-nul" ? fg::TranscodingSplitter::Nul : fg::TranscodingSplitter::Syslen, ec,
                                    argc > 4 ? (size_t)atoi(argv[4]) : (8u << 20));
        std::ifstream pin(argv[3], std::ios::binary);
        tsp.run(pin, *clone, std::cout, std::cerr);
        std::cout.flush();
        return 0;
    }
    if (fr == "gpu-line" || fr == "gpu-nul") {  // framing + UTF-8 validation on the GPU as well
        fg::GpuFramingSplitter gsp(fr == "gpu-line" ? fg::GpuFramingSplitter::Line : fg::GpuFramingSplitter::Nul,
                                   argc > 4 ? (size_t)atoi(argv[4]) : (8u << 20));
        std::ifstream gin(argv[3], std::ios::binary);
        gsp.run(gin, *clone, hex_sink, std::cerr);
        return 0;
    }
    fg::BatchingSplitter sp(fr == "line" ? fg::BatchingSplitter::Line : fr == "nul" ? fg::BatchingSplitter::Nul : fg::BatchingSplitter::Syslen,
                            argc > 4 ? (size_t)atoi(argv[4]) : 1000);
    std::ifstream in(argv[3], std::ios::binary);
    sp.run(in, *clone, [](fg::Record&& r) {
        fg::DecodeResult d;
        d.record = std::move(r);
        std::string c = fg::to_canonical(d);
        for (unsigned char ch : c) printf("%02x", ch);
        printf("\n");
    }, std::cerr);
    // the trait's per-line method
    auto one = dec->decode("<23>1 2015-08-05T15:53:45.637824Z testhostname appname 69 42 - hi");
    if (fmt == "rfc5424" && !(one.ok() && one.record.ts == 1438790025.637824 && *one.record.msg == "hi")) return 3;
    return 0;
}
