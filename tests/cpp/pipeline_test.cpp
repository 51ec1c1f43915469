// Two extractor handles gated against each other through the plain C ABI (the INTEGRATION.md snippet, compiled and run):
// results must equal those of un-gated calls, over several rounds.  Host side uses the HIP runtime API only.
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "myslam_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define MK(x) do { int r_ = (x); if (r_ != MYSLAM_OK) { printf("myslam error %d at line %d\n", r_, __LINE__); return 1; } } while (0)

static void make_image(std::vector<uint8_t>& img, int h, int w, unsigned seed) {          // blocky texture + noise: plenty of corners
    unsigned s = seed * 2654435761u + 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    std::vector<uint8_t> blocks((h / 8 + 1) * (w / 8 + 1));
    for (auto& b : blocks) b = (uint8_t)(rnd() % 200 + 20);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)y * w + x] = (uint8_t)(blocks[(y / 8) * (w / 8 + 1) + x / 8] + rnd() % 12);
}

int main() {
    const int B = 3, H = 240, W = 376, NF = 500;
    myslam_orb* orb[2];
    hipStream_t st[2];
    hipEvent_t fast[2];
    uint8_t* d_img[2]; myslam_keypoint* d_kp[2]; uint8_t* d_desc[2]; int32_t *d_cnt[2], *d_stat[2];
    int cap = 0;
    for (int s = 0; s < 2; s++) {
        MK(myslam_orb_create(&orb[s], NF, 1.2f, 8, 20, 7));
        CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&fast[s], hipEventDisableTiming));
        MK(myslam_orb_set_stream(orb[s], st[s]));
        cap = myslam_orb_max_keypoints_for(orb[s], H, W);
        if (cap <= 0) { printf("bad capacity %d\n", cap); return 1; }
        std::vector<uint8_t> img((size_t)B * H * W), one((size_t)H * W);
        for (int b = 0; b < B; b++) { make_image(one, H, W, 100 * s + b); memcpy(&img[(size_t)b * H * W], one.data(), one.size()); }
        CK(hipMalloc(&d_img[s], img.size())); CK(hipMemcpy(d_img[s], img.data(), img.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&d_kp[s], sizeof(myslam_keypoint) * B * cap)); CK(hipMalloc(&d_desc[s], (size_t)B * cap * 32));
        CK(hipMalloc(&d_cnt[s], 4 * B)); CK(hipMalloc(&d_stat[s], 4 * B));
    }
    auto run = [&](int s) {
        return myslam_orb_detect_and_compute_batch(orb[s], d_img[s], B, H, W, W, (size_t)H * W, nullptr, d_kp[s], d_desc[s], d_cnt[s], d_stat[s], cap);
    };
    auto fetch = [&](int s, std::vector<uint8_t>& out) -> int {
        std::vector<int32_t> cnt(B), stt(B);
        CK(hipMemcpy(cnt.data(), d_cnt[s], 4 * B, hipMemcpyDeviceToHost)); CK(hipMemcpy(stt.data(), d_stat[s], 4 * B, hipMemcpyDeviceToHost));
        std::vector<myslam_keypoint> kp((size_t)B * cap); std::vector<uint8_t> de((size_t)B * cap * 32);
        CK(hipMemcpy(kp.data(), d_kp[s], sizeof(myslam_keypoint) * kp.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(de.data(), d_desc[s], de.size(), hipMemcpyDeviceToHost));
        out.clear();
        for (int b = 0; b < B; b++) {
            if (stt[b] != 0 || cnt[b] < 50) { printf("image %d: status %d, %d key-points\n", b, stt[b], cnt[b]); return 1; }
            const uint8_t* k = reinterpret_cast<const uint8_t*>(&kp[(size_t)b * cap]);
            out.insert(out.end(), k, k + sizeof(myslam_keypoint) * cnt[b]);
            out.insert(out.end(), &de[(size_t)b * cap * 32], &de[(size_t)b * cap * 32] + 32 * cnt[b]);
        }
        return 0;
    };
    // reference: plain calls
    std::vector<uint8_t> ref[2], got;
    for (int s = 0; s < 2; s++) { MK(run(s)); CK(hipStreamSynchronize(st[s])); if (fetch(s, ref[s])) return 1; }
    // gated: each handle records its event after FAST and waits for the other's before FAST
    for (int s = 0; s < 2; s++) { MK(myslam_orb_set_fast_event(orb[s], fast[s])); MK(myslam_orb_set_fast_gate(orb[s], fast[1 - s])); }
    CK(hipEventRecord(fast[1], st[1]));                 // opens the first gate of handle 0
    for (int round = 0; round < 4; round++) {
        for (int s = 0; s < 2; s++) { CK(hipMemsetAsync(d_kp[s], 0, sizeof(myslam_keypoint) * B * cap, st[s])); CK(hipMemsetAsync(d_desc[s], 0, (size_t)B * cap * 32, st[s])); }
        MK(run(0)); MK(run(1));                          // always in this order
        CK(hipDeviceSynchronize());
        for (int s = 0; s < 2; s++) {
            if (fetch(s, got)) return 1;
            if (got != ref[s]) { printf("round %d handle %d: gated result differs\n", round, s); return 1; }
        }
    }
    for (int s = 0; s < 2; s++) { MK(myslam_orb_set_fast_event(orb[s], nullptr)); MK(myslam_orb_set_fast_gate(orb[s], nullptr)); MK(run(s)); }
    CK(hipDeviceSynchronize());
    for (int s = 0; s < 2; s++) { if (fetch(s, got)) return 1; if (got != ref[s]) { printf("un-gated again: differs\n"); return 1; } }
    for (int s = 0; s < 2; s++) { MK(myslam_orb_destroy(orb[s])); CK(hipStreamDestroy(st[s])); CK(hipEventDestroy(fast[s])); }
    printf("PIPELINE TEST OK (%zu + %zu result bytes per round)\n", ref[0].size(), ref[1].size());
    return 0;
}
