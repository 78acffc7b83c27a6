// enc_bench.cpp -- host-only timing of the read-level stagers: encode_delta (3-bit words, compare with reference bytes) against
// encode_planes (bit planes, XOR with the 2-bit reference plane), same synthetic batch, outputs compared byte for byte.
// build: g++ -O3 -std=c++17 -I include tools/src/enc_bench.cpp instrain_amd/csrc/seg_encode.o instrain_amd/csrc/obs_encode.o -lpthread -o tools/bin/enc_bench
// usage: enc_bench [threads] [n_pos] [depth] [p_skip] [p_mismatch] [p_N]      (ISX_ENC_PIN=<numa node>: spread the threads over the L3 domains of that node)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <algorithm>
#include "../../instrain_amd/csrc/seg_encode.h"

void isx_set_error(const std::string &m) { fprintf(stderr, "error: %s\n", m.c_str()); }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    const int64_t n_pos = argc > 2 ? atoll(argv[2]) : 100000000;
    const double depth = argc > 3 ? atof(argv[3]) : 3.0, p_skip = argc > 4 ? atof(argv[4]) : 0.10, p_mm = argc > 5 ? atof(argv[5]) : 0.003, p_n = argc > 6 ? atof(argv[6]) : 0.0;
    const int64_t n_seg = (int64_t)(depth * n_pos / 150.0);
    std::mt19937_64 rng(7);
    std::vector<uint8_t> ref((size_t)n_pos);
    for (auto &c : ref) c = (uint8_t)(rng() & 3);
    if (p_n > 0) for (int64_t i = 0; i < n_pos; i++) if ((rng() >> 11) * (1.0 / 9007199254740992.0) < p_n) ref[(size_t)i] = 4;
    std::vector<uint32_t> gpos((size_t)n_seg), pair((size_t)n_seg), bases((size_t)n_seg * 15 + 1);
    std::vector<uint8_t> len((size_t)n_seg);
    for (auto &g : gpos) g = (uint32_t)(rng() % (uint64_t)(n_pos - 150));
    std::sort(gpos.begin(), gpos.end());
    const uint64_t thr_skip = (uint64_t)(p_skip * 4294967296.0), thr_mm = (uint64_t)(p_mm * 4294967296.0);
    for (int64_t s = 0; s < n_seg; s++) {
        len[(size_t)s] = (s % 97 == 0) ? (uint8_t)(1 + rng() % 150) : (uint8_t)150;
        pair[(size_t)s] = (uint32_t)(s >> 1);
        uint32_t *w = &bases[(size_t)s * 15];
        for (int k = 0; k < 15; k++) w[k] = 0x24924924u;
        const bool clean = (rng() & 7) == 0;        // some reads without a skipped column (dual records)
        for (int j = 0; j < len[(size_t)s]; j++) {
            const uint64_t r = rng();
            uint32_t c = ref[(size_t)gpos[(size_t)s] + j];
            if (c > 3) c = (uint32_t)(r >> 60) & 3;
            if ((uint32_t)r < thr_mm) c = (c + 1 + ((r >> 40) % 3)) & 3;
            if (!clean && (uint32_t)(r >> 32) < thr_skip) c = 4 + ((r >> 50) & 1);
            w[j / 10] = (w[j / 10] & ~(7u << (3 * (j % 10)))) | (c << (3 * (j % 10)));
        }
    }
    isx_segs segs{n_seg, gpos.data(), len.data(), nullptr, pair.data(), bases.data()};
    std::vector<uint64_t> planes_raw((size_t)n_seg * 8 + 8);
    uint64_t *planes = reinterpret_cast<uint64_t *>((reinterpret_cast<uintptr_t>(planes_raw.data()) + 63) & ~(uintptr_t)63);
    double t0 = now();
    isx_planes_from_segs(&segs, T, planes);
    printf("planes_from_segs %.1f ms\n", (now() - t0) * 1e3);
    std::vector<uint8_t> p2((size_t)(n_pos + 3) / 4 + 64), pn((size_t)(n_pos + 7) / 8 + 64);
    int32_t has_n = 0;
    for (int rep = 0; rep < 3; rep++) {
        t0 = now();
        isx_pack_ref_planes(ref.data(), n_pos, T, p2.data(), pn.data(), &has_n);
        printf("pack_ref_planes %.2f ms (has_n %d)\n", (now() - t0) * 1e3, has_n);
    }
    isx_read_planes rp{n_seg, gpos.data(), len.data(), pair.data(), planes};
    isx_ref_planes rf{p2.data(), has_n ? pn.data() : nullptr};
    const char *pin_env = getenv("ISX_ENC_PIN");
    isxenc::HostPool pool(T, pin_env ? atoi(pin_env) : -1, pin_env != nullptr);
    if (pin_env) printf("threads pinned over the L3 domains of NUMA node %s\n", pin_env);
    int64_t slack = 8;
    const int64_t cap = isxenc::delta_groups_needed(pool, gpos.data(), n_seg, 64) * ISX_DREC_GROUP;
    std::vector<uint32_t> recA_raw((size_t)cap * 8 + 16), recB_raw((size_t)cap * 8 + 16), gbA((size_t)cap / 32), gbB((size_t)cap / 32);
    uint32_t *recA = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(recA_raw.data()) + 63) & ~(uintptr_t)63);
    uint32_t *recB = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(recB_raw.data()) + 63) & ~(uintptr_t)63);
    memset(recA, 0, (size_t)cap * 32); memset(recB, 0, (size_t)cap * 32);
    std::vector<uint32_t> cmin((size_t)cap / 32), cmax((size_t)cap / 32);
    std::vector<uint8_t> cany((size_t)cap / 32);
    int64_t nrA = 0, nrB = 0;
    for (int rep = 0; rep < 5; rep++) {
        for (int which = 0; which < 2; which++) {
            isxenc::SegJob J;
            J.n_seg = n_seg; J.n_pos = n_pos; J.n_mm_bins = 1; J.slack_groups = slack;
            J.cmin = cmin.data(); J.cmax = cmax.data(); J.cany = cany.data(); J.cap_rec = cap;
            int rc;
            t0 = now();
            if (which == 0) { J.in = segs; J.ref = ref.data(); J.rec = recA; J.gbase = gbA.data(); rc = isxenc::encode_delta(pool, J); nrA = J.n_rec; }
            else { J.in2 = rp; J.ref2 = rf.plane2; J.refn = rf.nplane; J.rec = recB; J.gbase = gbB.data(); rc = isxenc::encode_planes(pool, J); nrB = J.n_rec; }
            const double dt = now() - t0;
            printf("%s rc %d: %.2f ms, %.1f ns/seg/thread, %lld records (%lld pieces), need_slack %lld\n", which ? "encode_planes" : "encode_delta ", rc, dt * 1e3, dt * 1e9 * T / n_seg,
                   (long long)J.n_rec, (long long)J.n_pieces, (long long)J.need_slack);
            if (rc == isxenc::SEG_CAPACITY) slack = std::max(slack, J.need_slack);
        }
    }
    if (nrA != nrB) { printf("MISMATCH n_rec %lld vs %lld\n", (long long)nrA, (long long)nrB); return 1; }
    if (memcmp(recA, recB, (size_t)nrA * 32) != 0 || memcmp(gbA.data(), gbB.data(), (size_t)nrA / 32 * 4) != 0) {
        for (int64_t i = 0; i < nrA * 8; i++) if (recA[i] != recB[i]) { printf("MISMATCH at record %lld word %lld: %08x vs %08x\n", (long long)(i / 8), (long long)(i % 8), recA[i], recB[i]); break; }
        return 1;
    }
    printf("records identical: %lld records, %lld segments\n", (long long)nrA, (long long)n_seg);
    return 0;
}
