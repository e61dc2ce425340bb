// STFT analysis / synthesis on the GPU (hipFFT), so that time-domain audio can go in and out next to the solver.
//
// Replaces, in the reference's drivers, the third-party calls
//     X = pra.transform.analysis(mics_signals.T, framesize, framesize // 2, win=win_a)     overiva_oneshot.py:293-295
//     y = pra.transform.synthesis(Y, framesize, framesize // 2, win=win_s)                 overiva_oneshot.py:371-379
// (pyroomacoustics 0.1.23, source absent from /root/reference: PARITY UNPINNED; the framing below restates its
// block-processing convention -- every frame consumes `hop` new samples behind `frame - hop` old ones, the state
// before the first sample is zero, so n_frames = n_samples / hop -- and is pinned against scipy.signal.stft and
// the NumPy oracle oracle/stft_oracle.py).
//
//   analysis : x (n_samples, M) float32  ->  X (T, F, M) complex64,  F = frame/2 + 1
//              frame t = win_a * [x[t*hop - (frame-hop) .. t*hop + hop)], rfft  (one batched R2C of T*M transforms)
//   synthesis: Y (T, F, K) complex64     ->  y (T*hop, K) float32
//              irfft of every (t, k), * win_s, overlap-add; the first frame-hop samples of frame 0 (the zero state)
//              are dropped
// Kernels: framing + window (coalesced over samples), (T*M, F) <-> (T, F, M) transposes as gathers,
// overlap-add as a gather (every output sample sums its frame/hop contributions in frame order: deterministic).
#include <hipfft/hipfft.h>

#include <algorithm>
#include <string>

#include "oiva_internal.h"

using namespace oiva;

namespace {

int sfail(int code, const std::string& msg) { return oiva::fail_with(code, msg); }   // oiva_last_error() reports it
#define S_HIP(expr)                                                                                       \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return sfail(OIVA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define S_FFT(expr)                                                                                  \
    do {                                                                                             \
        hipfftResult r_ = (expr);                                                                    \
        if (r_ != HIPFFT_SUCCESS) return sfail(OIVA_ERR_HIP, std::string(#expr) + ": hipfft error " + std::to_string((int)r_)); \
    } while (0)
#define S_NEED(cond, code, msg)               \
    do {                                      \
        if (!(cond)) return sfail(code, msg); \
    } while (0)

// frames[(t*C + c)*L + n] = win[n] * x[(t*hop - (L-hop) + n), c]   (zero before the first sample)
__global__ __launch_bounds__(kBlock) void frame_kernel(const float* __restrict__ x, const float* __restrict__ win,
                                                       float* __restrict__ frames, int n_samples, int C, int L, int hop,
                                                       int T) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (e >= (long long)T * C * L) return;
    const int n = (int)(e % L);
    const long long tc = e / L;
    const int c = (int)(tc % C), t = (int)(tc / C);
    const long long s = (long long)t * hop - (L - hop) + n;
    const float v = (s >= 0 && s < n_samples) ? x[s * C + c] : 0.f;
    frames[e] = v * (win ? win[n] : 1.f);
}

// spec (T*C, F) complex  <->  X (T, F, C) complex
__global__ __launch_bounds__(kBlock) void to_tfc_kernel(const float2* __restrict__ spec, float2* __restrict__ X, int T, int F,
                                                        int C) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;      // over X: ((t*F + f)*C + c)
    if (e >= (long long)T * F * C) return;
    const int c = (int)(e % C);
    const long long tf = e / C;
    const int f = (int)(tf % F), t = (int)(tf / F);
    X[e] = spec[((long long)t * C + c) * F + f];
}
__global__ __launch_bounds__(kBlock) void from_tfc_kernel(const float2* __restrict__ Y, float2* __restrict__ spec, int T, int F,
                                                          int C) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;      // over spec: ((t*C + c)*F + f)
    if (e >= (long long)T * F * C) return;
    const int f = (int)(e % F);
    const long long tc = e / F;
    const int c = (int)(tc % C), t = (int)(tc / C);
    spec[e] = Y[((long long)t * F + f) * C + c];
}

// y[s, c] = (1/L) * sum over the frames t that cover sample s of win[n] * frames[(t*C + c)*L + n],
// n = s + (L-hop) - t*hop; frames in increasing t (fixed order)
__global__ __launch_bounds__(kBlock) void overlap_add_kernel(const float* __restrict__ frames, const float* __restrict__ win,
                                                             float* __restrict__ y, int C, int L, int hop, int T) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;      // over y: s*C + c
    const long long n_out = (long long)T * hop;
    if (e >= n_out * C) return;
    const int c = (int)(e % C);
    const long long s = e / C;
    const long long pos = s + (L - hop);                 // position on the axis that includes the zero state
    long long t_hi = pos / hop;                          // last frame covering pos
    if (t_hi > T - 1) t_hi = T - 1;
    long long t_lo = (pos - L) / hop + 1;                // first frame with t*hop + L > pos
    if (pos - L < 0) t_lo = 0;
    float acc = 0.f;
    for (long long t = t_lo; t <= t_hi; ++t) {
        const int n = (int)(pos - t * hop);
        if (n >= 0 && n < L) acc += frames[((long long)t * C + c) * L + n] * (win ? win[n] : 1.f);
    }
    y[e] = acc * (1.f / (float)L);
}

}  // namespace

struct oiva_stft {
    int device = 0;
    int n_samples = 0, C = 0, L = 0, hop = 0, T = 0, F = 0;
    hipStream_t stream = nullptr;
    hipfftHandle fwd = 0, inv = 0;
    bool have_fwd = false, have_inv = false;
    int inv_chan = -1;        // channel count the inverse plan was made for
    float* win_a = nullptr;   // (L) or nullptr
    float* win_s = nullptr;
    float* x = nullptr;       // (n_samples, C) | (T*hop, C)
    float* frames = nullptr;  // (T*C, L)
    float2* spec = nullptr;   // (T*C, F)
    float2* X = nullptr;      // (T, F, C)
};

extern "C" {

int oiva_stft_create(oiva_stft** out, int device, int n_samples, int n_chan, int frame, int hop, const float* win_a,
                     const float* win_s) {
    S_NEED(out != nullptr, OIVA_ERR_ARG, "null out pointer");
    *out = nullptr;
    S_NEED(n_chan >= 1 && frame >= 2 && frame % 2 == 0, OIVA_ERR_ARG, "frame must be even, n_chan >= 1");
    S_NEED(hop >= 1 && hop <= frame, OIVA_ERR_ARG, "hop must be in 1..frame");
    S_NEED(n_samples >= hop, OIVA_ERR_ARG, "fewer samples than one hop");
    int ndev = 0;
    S_HIP(hipGetDeviceCount(&ndev));
    S_NEED(device >= 0 && device < ndev, OIVA_ERR_ARG, "no such device");
    S_HIP(hipSetDevice(device));
    oiva_stft* p = new oiva_stft();
    p->device = device;
    p->n_samples = n_samples;
    p->C = n_chan;
    p->L = frame;
    p->hop = hop;
    p->T = n_samples / hop;
    p->F = frame / 2 + 1;
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    const size_t nx = (size_t)std::max(n_samples, p->T * hop) * n_chan;
    auto alloc = [&](void** ptr, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(ptr, bytes);
    };
    alloc((void**)&p->x, nx * sizeof(float));
    alloc((void**)&p->frames, (size_t)p->T * n_chan * frame * sizeof(float));
    alloc((void**)&p->spec, (size_t)p->T * n_chan * p->F * sizeof(float2));
    alloc((void**)&p->X, (size_t)p->T * n_chan * p->F * sizeof(float2));
    if (win_a) {
        alloc((void**)&p->win_a, frame * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(p->win_a, win_a, frame * sizeof(float), hipMemcpyHostToDevice);
    }
    if (win_s) {
        alloc((void**)&p->win_s, frame * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(p->win_s, win_s, frame * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        oiva_stft_destroy(p);
        return sfail(OIVA_ERR_HIP, std::string("allocation failed: ") + hipGetErrorString(e));
    }
    *out = p;
    return OIVA_OK;
}

int oiva_stft_destroy(oiva_stft* p) {
    if (!p) return OIVA_OK;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->have_fwd) (void)hipfftDestroy(p->fwd);
    if (p->have_inv) (void)hipfftDestroy(p->inv);
    void* bufs[] = {p->win_a, p->win_s, p->x, p->frames, p->spec, p->X};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return OIVA_OK;
}

int oiva_stft_shape(oiva_stft* p, int* n_frames, int* n_freq) {
    S_NEED(p, OIVA_ERR_ARG, "null handle");
    if (n_frames) *n_frames = p->T;
    if (n_freq) *n_freq = p->F;
    return OIVA_OK;
}

int oiva_stft_analysis(oiva_stft* p, const float* x_host, void* X_host, void** X_dev) {
    S_NEED(p && x_host, OIVA_ERR_ARG, "null argument");
    S_HIP(hipSetDevice(p->device));
    const int T = p->T, C = p->C, L = p->L, F = p->F;
    if (!p->have_fwd) {
        int n[1] = {L};
        S_FFT(hipfftPlanMany(&p->fwd, 1, n, nullptr, 1, L, nullptr, 1, F, HIPFFT_R2C, T * C));
        S_FFT(hipfftSetStream(p->fwd, p->stream));
        p->have_fwd = true;
    }
    S_HIP(hipMemcpyAsync(p->x, x_host, (size_t)p->n_samples * C * sizeof(float), hipMemcpyHostToDevice, p->stream));
    const long long ne = (long long)T * C * L;
    hipLaunchKernelGGL(frame_kernel, dim3((unsigned)((ne + kBlock - 1) / kBlock)), dim3(kBlock), 0, p->stream, p->x, p->win_a,
                       p->frames, p->n_samples, C, L, p->hop, T);
    S_HIP(hipGetLastError());
    S_FFT(hipfftExecR2C(p->fwd, p->frames, reinterpret_cast<hipfftComplex*>(p->spec)));
    const long long nx = (long long)T * F * C;
    hipLaunchKernelGGL(to_tfc_kernel, dim3((unsigned)((nx + kBlock - 1) / kBlock)), dim3(kBlock), 0, p->stream, p->spec, p->X, T, F, C);
    S_HIP(hipGetLastError());
    if (X_host) S_HIP(hipMemcpyAsync(X_host, p->X, (size_t)nx * sizeof(float2), hipMemcpyDeviceToHost, p->stream));
    S_HIP(hipStreamSynchronize(p->stream));
    if (X_dev) *X_dev = p->X;
    return OIVA_OK;
}

int oiva_stft_synthesis(oiva_stft* p, const void* Y_host, int n_chan, float* y_host) {
    S_NEED(p && Y_host && y_host, OIVA_ERR_ARG, "null argument");
    S_NEED(n_chan >= 1 && n_chan <= p->C, OIVA_ERR_ARG, "synthesis takes at most the channel count of the plan");
    S_HIP(hipSetDevice(p->device));
    const int T = p->T, C = n_chan, L = p->L, F = p->F;
    // one inverse plan per channel count (the solver returns fewer channels than the analysis had)
    if (!p->have_inv || p->inv_chan != C) {
        if (p->have_inv) S_FFT(hipfftDestroy(p->inv));
        int n[1] = {L};
        S_FFT(hipfftPlanMany(&p->inv, 1, n, nullptr, 1, F, nullptr, 1, L, HIPFFT_C2R, T * C));
        S_FFT(hipfftSetStream(p->inv, p->stream));
        p->have_inv = true;
        p->inv_chan = C;
    }
    const long long nx = (long long)T * F * C;
    S_HIP(hipMemcpyAsync(p->X, Y_host, (size_t)nx * sizeof(float2), hipMemcpyHostToDevice, p->stream));
    hipLaunchKernelGGL(from_tfc_kernel, dim3((unsigned)((nx + kBlock - 1) / kBlock)), dim3(kBlock), 0, p->stream, p->X, p->spec, T, F, C);
    S_HIP(hipGetLastError());
    S_FFT(hipfftExecC2R(p->inv, reinterpret_cast<hipfftComplex*>(p->spec), p->frames));
    const long long ny = (long long)T * p->hop * C;
    hipLaunchKernelGGL(overlap_add_kernel, dim3((unsigned)((ny + kBlock - 1) / kBlock)), dim3(kBlock), 0, p->stream, p->frames,
                       p->win_s, p->x, C, L, p->hop, T);
    S_HIP(hipGetLastError());
    S_HIP(hipMemcpyAsync(y_host, p->x, (size_t)ny * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    S_HIP(hipStreamSynchronize(p->stream));
    return OIVA_OK;
}

}  // extern "C"
