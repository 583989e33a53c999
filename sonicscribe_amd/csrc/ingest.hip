// Device-resident ingest.  The reference keeps every 2048-byte WebSocket chunk in a host dict
// (backend/audio_manager.py:21-33), concatenates the chunks of a speech segment on the host for every partial / final decode
// (:106-123), converts int16 -> float32 / 32768 (backend/transcription_manager.py:45-54), peak-normalises and quantises back to PCM_16
// (backend/asr.py:247-276) before the feature extractor sees it.  Here the raw wire samples of a session live in a ring in HBM and a
// decode stages its windows from the ring with that arithmetic done on the device:
//     f = s / 32768            (exact in fp32)
//     m = max |f| over the REQUEST (all its 30 s windows)   = max |s| / 32768, so the peak is reduced on the integers
//     m > 1e-6 (i.e. max |s| >= 1):  w = f / m  (fp32 division, round to nearest)   else  w = f
//     q = clip(rint(w * 32767))  (round-half-even, as lrintf)  -> the int16 PCM the log-mel kernel loads
// Bit-identical with sonicscribe_amd/frontend.py normalise_to_int16(pcm_bytes_to_float(bytes)) by construction (IEEE division and
// multiplication, no contraction), which the GPU tests check on the staged samples' downstream results.
// HBM-bound byte work: 2 B in + 2 B out per sample; 16 B per lane.
#include "common.h"
#include "kernels.h"

// peak[req] = max |s| over every ring window of the request (ordered ints: atomicMax on int)
__global__ __launch_bounds__(256) void ring_peak_kernel(RingStageArgs a) {
    const int w = blockIdx.y;
    const short* ring = a.ring[w];
    if (!ring) return;                                   // host window: already normalised
    const long cap = a.ring_cap[w], start = a.start[w];  // start: position in the ring (already reduced modulo cap), n <= cap
    const int n = a.n[w];
    int mx = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        long p = start + i; p = p >= cap ? p - cap : p;
        const int s = ring[p];
        mx = max(mx, s < 0 ? -s : s);
    }
    mx = (int)wave_max((float)mx);                       // |s| <= 32768: exact in fp32
    if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(a.peak + a.req_of[w], mx);
}

__global__ __launch_bounds__(256) void ring_stage_kernel(RingStageArgs a) {
    const int w = blockIdx.y;
    const short* ring = a.ring[w];
    if (!ring) return;
    const long cap = a.ring_cap[w], start = a.start[w];
    const int n = a.n[w];
    const int pk = a.peak[a.req_of[w]];
    const float m = (float)pk * (1.0f / 32768.0f);       // exact: pk * 2^-15
    short* dst = a.pcm + (long)w * a.win_cap;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        long p = start + i; p = p >= cap ? p - cap : p;
        const float f = (float)ring[p] * (1.0f / 32768.0f);
        const float v = pk > 0 ? __fdiv_rn(f, m) : f;    // m > 1e-6  <=>  pk >= 1
        const float q = rintf(__fmul_rn(v, 32767.0f));
        dst[i] = (short)fminf(fmaxf(q, -32768.f), 32767.f);
    }
}

void launch_ring_stage(const RingStageArgs& a, int W, int max_n, hipStream_t s) {
    if (W < 1 || max_n < 1) return;
    const int bx = (max_n + 4095) / 4096 < 1 ? 1 : (max_n + 4095) / 4096;   // 16 samples per thread
    hipLaunchKernelGGL(ring_peak_kernel, dim3(bx, W), dim3(256), 0, s, a);
    hipLaunchKernelGGL(ring_stage_kernel, dim3(bx, W), dim3(256), 0, s, a);
}
