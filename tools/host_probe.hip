// host_probe.hip -- what the GPU box's host can feed the device with (tools only, not part of the library):
//   * multi-threaded host copy bandwidth (the encode pass of the streaming pipeline is a streaming copy)
//   * pinned H2D / D2H hipMemcpyAsync bandwidth, alone and duplex
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/host_probe tools/host_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t N = (size_t)1 << 30;           // 1 GiB source
    uint8_t *src = (uint8_t *)aligned_alloc(4096, N), *dst = (uint8_t *)aligned_alloc(4096, N / 4);
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    for (unsigned nt : {1u, 4u, 8u, 16u, 32u, 64u, 96u, 128u}) {
        if (nt > std::thread::hardware_concurrency()) break;
        // first touch by the threads that use it
        auto run = [&](bool touch) {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([&, t] {
                    const size_t a = N / nt * t, b = N / nt * (t + 1);
                    if (touch) { memset(src + a, 1, b - a); memset(dst + a / 4, 1, (b - a) / 4); return; }
                    // read 4 bytes, write 1: the shape of the 8-byte -> 2-byte record encode
                    const uint64_t *s = (const uint64_t *)(src + a);
                    uint16_t *d = (uint16_t *)(dst + a / 4);
                    const size_t n = (b - a) / 8;
                    for (size_t i = 0; i < n; i++) d[i] = (uint16_t)(s[i] ^ (s[i] >> 32));
                });
            for (auto &x : th) x.join();
        };
        run(true);
        double best = 1e9;
        for (int r = 0; r < 3; r++) { const double t0 = now(); run(false); best = std::min(best, now() - t0); }
        printf("host encode-shaped copy  threads %3u  read %.1f GB/s (+ write %.1f GB/s)\n", nt, N / best / 1e9, N / 4 / best / 1e9);
    }
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) { printf("no device\n"); return 0; }
    void *pin_a, *pin_b, *da, *db;
    const size_t B = (size_t)256 << 20;
    hipHostMalloc(&pin_a, B, hipHostMallocDefault); hipHostMalloc(&pin_b, B, hipHostMallocDefault);
    memset(pin_a, 1, B); memset(pin_b, 2, B);
    hipMalloc(&da, B); hipMalloc(&db, B);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int mode = 0; mode < 3; mode++) {
        double best = 1e9;
        for (int r = 0; r < 4; r++) {
            hipDeviceSynchronize();
            const double t0 = now();
            if (mode == 0 || mode == 2) hipMemcpyAsync(da, pin_a, B, hipMemcpyHostToDevice, s1);
            if (mode == 1 || mode == 2) hipMemcpyAsync(pin_b, db, B, hipMemcpyDeviceToHost, s2);
            hipStreamSynchronize(s1); hipStreamSynchronize(s2);
            best = std::min(best, now() - t0);
        }
        printf("%s 256 MiB: %.2f ms = %.1f GB/s per direction\n", mode == 0 ? "H2D" : mode == 1 ? "D2H" : "duplex", best * 1e3, B / best / 1e9);
    }
    // small pieces: 16 MiB chunks back to back on one stream
    {
        const size_t P = (size_t)16 << 20;
        hipDeviceSynchronize();
        const double t0 = now();
        for (size_t o = 0; o < B; o += P) hipMemcpyAsync((char *)da + o, (char *)pin_a + o, P, hipMemcpyHostToDevice, s1);
        hipStreamSynchronize(s1);
        printf("H2D 16 x 16 MiB: %.1f GB/s\n", B / (now() - t0) / 1e9);
    }
    return 0;
}
