// In-kernel timeline of logmel_power_kernel (thread 0 of every block stamps the 100 MHz wall clock at the phase boundaries):
//   hipcc --offload-arch=gfx950 -O3 -DLM_TRACE -I sonicscribe_amd/csrc tools/mel_timeline.hip -o /tmp/mel_timeline && /tmp/mel_timeline
// phases: 0 entry, 1 PCM + tables in LDS, 2 fold done, 3 DFT (MFMA loop) done, 4 power spectrum in LDS, 5 mel + log10 written
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../sonicscribe_amd/csrc/logmel.hip"
thread_local LaunchOpts g_opts;
void launch_fill_i32(int*, int, int, hipStream_t) {}      // (launch_logmel's helper; the tool launches the kernel itself)

int main() {
    const int B = 32, n_frames = 3000, n_mels = 128, n = 320000;
    int16_t* pcm; int* ns; float *logspec, *win, *ct, *st, *w; int *segmax, *lo, *cnt, *off; long long* tr;
    hipMalloc(&pcm, (size_t)B * 480000 * 2); hipMalloc(&ns, B * 4); hipMalloc(&logspec, (size_t)B * n_frames * n_mels * 4); hipMalloc(&segmax, B * 4);
    std::vector<int16_t> h((size_t)B * 480000);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (int16_t)(((i * 2654435761u) >> 16) & 0x3FFF) - 8192;
    hipMemcpy(pcm, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::vector<int> hn(B, n); hipMemcpy(ns, hn.data(), B * 4, hipMemcpyHostToDevice);
    std::vector<float> hw(400), hc(400), hs(400), taps(128 * 16, 0.01f);
    for (int i = 0; i < 400; ++i) { hw[i] = 0.5f - 0.5f * cosf(2 * M_PI * i / 400); hc[i] = cosf(2 * M_PI * i / 400); hs[i] = sinf(2 * M_PI * i / 400); }
    std::vector<int> hlo(128), hcnt(128, 16), hoff(128);
    for (int m = 0; m < 128; ++m) { hlo[m] = m + (m > 64 ? (m - 64) / 2 : 0); hoff[m] = m * 16; }      // a bank of the real one's size (16 taps each)
    hipMalloc(&win, 1600); hipMalloc(&ct, 1600); hipMalloc(&st, 1600); hipMalloc(&w, taps.size() * 4); hipMalloc(&lo, 512); hipMalloc(&cnt, 512); hipMalloc(&off, 512);
    hipMemcpy(win, hw.data(), 1600, hipMemcpyHostToDevice); hipMemcpy(ct, hc.data(), 1600, hipMemcpyHostToDevice); hipMemcpy(st, hs.data(), 1600, hipMemcpyHostToDevice);
    hipMemcpy(w, taps.data(), taps.size() * 4, hipMemcpyHostToDevice); hipMemcpy(lo, hlo.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(cnt, hcnt.data(), 512, hipMemcpyHostToDevice); hipMemcpy(off, hoff.data(), 512, hipMemcpyHostToDevice);
    const int tiles = (2002 + 31) / 32;
    hipMalloc(&tr, (size_t)B * tiles * 8 * 8); hipMemset(tr, 0, (size_t)B * tiles * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(lm_trace), &tr, sizeof tr);
    LogmelConst lc{win, ct, st, lo, cnt, off, w};
    hipFuncSetAttribute((const void*)logmel_power_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LM_DYN_LDS);
    hipEvent_t a, b2; hipEventCreate(&a); hipEventCreate(&b2);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(logmel_power_kernel, dim3(tiles, B), dim3(256), LM_DYN_LDS, 0, pcm, 480000L, ns, lc, logspec, segmax, n_frames, n_mels);
        hipEventRecord(b2); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b2); printf("kernel %.1f us\n", ms * 1e3);
    }
    std::vector<long long> t((size_t)B * tiles * 8);
    hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost);
    const char* names[5] = {"PCM + tables -> LDS", "fold", "DFT (200 x mfma_f32_32x32x2)", "power spectrum", "mel bank + log10 + store"};
    long long first = t[0], last = 0;
    for (size_t blk = 0; blk < (size_t)B * tiles; ++blk) { first = std::min(first, t[blk * 8]); last = std::max(last, t[blk * 8 + 5]); }
    printf("%zu blocks, first entry -> last exit %.1f us\n", (size_t)B * tiles, (last - first) * 0.01);
    for (int p = 0; p < 5; ++p) {
        std::vector<double> d;
        for (size_t blk = 0; blk < (size_t)B * tiles; ++blk) d.push_back((t[blk * 8 + p + 1] - t[blk * 8 + p]) * 0.01);
        std::sort(d.begin(), d.end());
        printf("  %-32s median %6.2f us   p90 %6.2f us\n", names[p], d[d.size() / 2], d[d.size() * 9 / 10]);
    }
    std::vector<double> tot;
    for (size_t blk = 0; blk < (size_t)B * tiles; ++blk) tot.push_back((t[blk * 8 + 5] - t[blk * 8]) * 0.01);
    std::sort(tot.begin(), tot.end());
    printf("  block total median %.2f us\n", tot[tot.size() / 2]);
    return 0;
}
