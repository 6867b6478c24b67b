// Which host-path SHAPE moves uploads and downloads at the same time on this platform?  (the pipelined decode
// paths of round 3 run their two directions one after the other -- 57 GB/s for both together -- while fg_transcode_batch's two lanes get
// ~70 GB/s.)  Copies only, no kernels: 32 slices of 32 MiB up, and per slice `parts` downloads of `down` bytes in total.
//   shape A  three roles: all uploads on stream U (queued first), downloads on stream D behind an event per slice
//   shape B  two lanes: slice k on lane k & 1 -- upload, then its downloads, on the same stream
//   shape K / S  shape A with a kernel per slice on a third stream between upload and downloads (stream index 3): K = a kernel without
//            private (scratch) memory, S = the same kernel with ~512 B of scratch per lane (k_ltsv and k_gelf_general reserve 508 / 476 B,
//            k_rfc5424 none -- and only the RFC5424 corpora overlap their two directions through fg_decode_batch)
// usage: pipeline_shapes <A|B> <U> <D> <prime> <down_MiB_per_slice> <parts>     U, D, prime = stream indices 0..3 (creation order);
//        prime = the stream that copies a few bytes each way before anything else (-1: none)
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/pipeline_shapes.cpp -o tools/probe/pipeline_shapes
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <bool SCRATCH>
__global__ void touch(const uint8_t* in, uint8_t* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    if (SCRATCH) {
        uint32_t a[128];  // indexed by data: lives in private memory
        for (int j = 0; j < 128; ++j) a[j] = j * 2654435761u;
        for (size_t p = i * 16; p < n; p += stride * 16) {
            const uint32_t v = *(const uint32_t*)(in + p);
            a[v & 127] += v;
            acc += a[(v >> 8) & 127];
        }
    } else {
        for (size_t p = i * 16; p < n; p += stride * 16) acc += *(const uint32_t*)(in + p) * 2654435761u;
    }
    if (acc == 0x12345678u) out[i & 4095] = (uint8_t)acc;
}

int main(int argc, char** argv) {
    if (argc < 7) return 2;
    const char shape = argv[1][0];
    const int U = atoi(argv[2]), D = atoi(argv[3]), prime = atoi(argv[4]);
    const size_t up = 32u << 20, down = (size_t)(atof(argv[5]) * (1 << 20));
    const int parts = atoi(argv[6]), slices = 32;
    uint8_t *h_in, *h_out, *d_in, *d_out;
    hipHostMalloc((void**)&h_in, up * slices, hipHostMallocDefault);
    hipHostMalloc((void**)&h_out, down * slices + 4096, hipHostMallocDefault);
    hipMalloc((void**)&d_in, up * slices);
    hipMalloc((void**)&d_out, down * slices + 4096);
    memset(h_in, 1, up * slices);
    memset(h_out, 0, down * slices + 4096);
    hipStream_t s[4];
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    std::vector<hipEvent_t> ev(slices);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    std::vector<hipEvent_t> ev2(slices);
    for (auto& e : ev2) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (prime >= 0) {
        hipMemcpyAsync(d_in, h_in, 8, hipMemcpyHostToDevice, s[prime]);
        hipMemcpyAsync(h_out, d_out, 8, hipMemcpyDeviceToHost, s[prime]);
        hipStreamSynchronize(s[prime]);
    }
    double best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        const size_t part = down / parts;
        if (shape == 'K' || shape == 'S') {
            for (int k = 0; k < slices; ++k) {
                hipMemcpyAsync(d_in + k * up, h_in + k * up, up, hipMemcpyHostToDevice, s[U]);
                hipEventRecord(ev[k], s[U]);
            }
            for (int k = 0; k < slices; ++k) {
                hipStreamWaitEvent(s[3], ev[k], 0);
                if (shape == 'S') touch<true><<<2048, 64, 0, s[3]>>>(d_in + k * up, d_out + k * down, up);
                else touch<false><<<2048, 64, 0, s[3]>>>(d_in + k * up, d_out + k * down, up);
                hipEventRecord(ev2[k], s[3]);
                hipStreamWaitEvent(s[D], ev2[k], 0);
                for (int j = 0; j < parts; ++j)
                    hipMemcpyAsync(h_out + k * down + j * part, d_out + k * down + j * part, part, hipMemcpyDeviceToHost, s[D]);
            }
            hipStreamSynchronize(s[3]);
        } else if (shape == 'A') {
            for (int k = 0; k < slices; ++k) {
                hipMemcpyAsync(d_in + k * up, h_in + k * up, up, hipMemcpyHostToDevice, s[U]);
                hipEventRecord(ev[k], s[U]);
            }
            for (int k = 0; k < slices; ++k) {
                hipStreamWaitEvent(s[D], ev[k], 0);
                for (int j = 0; j < parts; ++j)
                    hipMemcpyAsync(h_out + k * down + j * part, d_out + k * down + j * part, part, hipMemcpyDeviceToHost, s[D]);
            }
        } else {
            for (int k = 0; k < slices; ++k) {
                hipStream_t l = s[(k & 1) ? D : U];
                hipMemcpyAsync(d_in + k * up, h_in + k * up, up, hipMemcpyHostToDevice, l);
                for (int j = 0; j < parts; ++j)
                    hipMemcpyAsync(h_out + k * down + j * part, d_out + k * down + j * part, part, hipMemcpyDeviceToHost, l);
            }
        }
        hipStreamSynchronize(s[U]);
        hipStreamSynchronize(s[D]);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rep && dt < best) best = dt;
    }
    printf("shape %c U=s%d D=s%d prime=%d down %.1f MiB x%d parts: %.2f ms, in %.1f GB/s, in+out %.1f GB/s\n", shape, U, D, prime,
           down / 1048576.0, parts, best * 1e3, up * slices / best / 1e9, (up + down) * slices / best / 1e9);
    return 0;
}
