// Exercises the C++ host mirror (flowgger_amd/host/fg_decoder.hpp): frames a file like the
// reference splitters, decodes in batches on the GPU, prints one hex canonical Record per Ok line
// on stdout and the reference's error lines on stderr.  usage: host_mirror_test <rfc5424|ltsv|gelf> <line|nul|syslen|gpu-line|gpu-nul|pipe-line|pipe-nul|pipe-syslen> <file> [batch]
//        host_mirror_test <fmt> <fd-line|fd-gpu-line|fd-pipe-line|fd-syslen> - <idle_timeout_ms> [max_latency_ms]   frames STDIN (a pipe / socket)
//            with the flush policy of fg::FlushPolicy: what has arrived is decoded when the source runs dry, the idle timeout closes
//        host_mirror_test <fmt> micro <file> <max_lines> [max_latency_ms]   the per-record callers' adapter: one push() per line of the file
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
    if (fr.rfind("fd-", 0) == 0) {  // a live source: stdin
        fg::FdSource src(0);
        fg::FlushPolicy pol;
        pol.idle_timeout_ms = argc > 4 ? atoi(argv[4]) : -1;
        if (argc > 5) pol.max_latency_ms = atoi(argv[5]);
        auto flush_sink = [&](fg::Record&& r) {
            hex_sink(std::move(r));
            fflush(stdout);
        };
        if (fr == "fd-line" || fr == "fd-syslen") {
            fg::BatchingSplitter sp(fr == "fd-line" ? fg::BatchingSplitter::Line : fg::BatchingSplitter::Syslen);
            sp.run(src, pol, *clone, flush_sink, std::cerr);
        } else if (fr == "fd-gpu-line") {
            fg::GpuFramingSplitter sp(fg::GpuFramingSplitter::Line);
            sp.run(src, pol, *clone, flush_sink, std::cerr);
        } else {
            fg::EncoderConfig ec;
            ec.encoder = FG_ENC_GELF;
            ec.merger = FG_MERGE_LINE;
            fg::TranscodingSplitter sp(fg::TranscodingSplitter::Line, ec);
            sp.run(src, pol, *clone, std::cout, std::cerr);
        }
        fflush(stdout);
        return 0;
    }
    if (fr == "micro") {  // udp_input.rs:78-88 with the micro-batching adapter: one record per push()
        fg::MicroBatcher mb(*clone, hex_sink, [](const char* e, std::string_view) { std::cerr << e << "\n"; },
                            argc > 4 ? (size_t)atoi(argv[4]) : 4096, argc > 5 ? atoi(argv[5]) : 5);
        std::ifstream min(argv[3], std::ios::binary);
        std::string rec;
        size_t pushed = 0, max_pending = 0;
        while (std::getline(min, rec, '\n')) {
            mb.push(rec);
            mb.poll();
            ++pushed;
            if (mb.pending() > max_pending) max_pending = mb.pending();
        }
        mb.flush();
        fprintf(stderr, "micro: %zu records, at most %zu parked\n", pushed, max_pending);
        return 0;
    }
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
