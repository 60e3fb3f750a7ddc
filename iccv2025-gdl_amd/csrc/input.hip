// input.hip -- the two device-side stages of the reference's input pipeline.
//
// 1. log-magnitude STFT of a clipped waveform, what every dataset of the reference computes per sample on the host:
//        resamples[resamples > 1.] = 1.; resamples[resamples < -1.] = -1.
//        spectrogram = librosa.stft(resamples, n_fft=512, hop_length=353)      (256 / 128 for Kinetics-Sounds, VGGSound)
//        spectrogram = np.log(np.abs(spectrogram) + 1e-7)
//    (/root/reference/dataset/CramedDataset.py:62-66, KSDataset.py:144-149, VGGSoundDataset.py:117-122).
//    librosa is a third-party dependency that is not vendored in the reference; its published algorithm (librosa.stft,
//    defaults win_length = n_fft, window = 'hann' (periodic: scipy.signal.get_window(..., fftbins=True)), center = True)
//    is: pad the signal by n_fft/2 on both sides (pad_mode 'constant' = zeros since librosa 0.10, 'reflect' before),
//    frame t = padded[t*hop .. t*hop + n_fft), X[k][t] = sum_n frame[n] * hann[n] * exp(-2 pi i k n / n_fft),
//    k = 0 .. n_fft/2, 1 + L/hop frames.  Output [B][n_fft/2+1][frames] float32, the tensor the DataLoader yields.
//
//    One block = 4 frames of one waveform, one thread per frequency bin: a direct DFT in fp32 against an exact
//    (double-precision-generated) twiddle table in LDS.  12 032 frames x 257 bins x 512 samples is 3.2 GMAC for a
//    B=64 CREMA-D batch -- microseconds of VALU time, so an FFT would buy nothing and the direct sum is the more
//    accurate of the two in fp32.
//
// 2. ToTensor() + Normalize(mean, std) of decoded frames (CramedDataset.py:77-81): uint8 HWC -> float32 CHW,
//    ((x / 255) - mean[c]) / std[c] in that order of fp32 operations (bit-identical to torchvision on the CPU).
//    Decoding, resizing and the random crops stay on the host (PIL), as in the reference.
#include "common.h"
#include "ops.h"

namespace gdl {

constexpr int LS_FPB = 4;  // frames per block

__global__ void logspec_kernel(const float* __restrict__ wave, float* __restrict__ out, int L, int n_fft, int hop, int frames,
                               int reflect) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ls_smem[];
    float2* tw = (float2*)ls_smem;               // [n_fft] (cos, sin)(2 pi n / n_fft)
    float* xw = (float*)(tw + n_fft);            // [LS_FPB][n_fft] windowed frames
    const int b = blockIdx.y, f0 = blockIdx.x * LS_FPB, bins = n_fft / 2 + 1, pad = n_fft / 2;
    for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
        double s, c;
        sincospi(2.0 * (double)n / (double)n_fft, &s, &c);
        tw[n] = make_float2((float)c, (float)s);
    }
    __syncthreads();
    const float* w = wave + (size_t)b * L;
    for (int i = threadIdx.x; i < LS_FPB * n_fft; i += blockDim.x) {
        const int j = i / n_fft, n = i - j * n_fft;
        int p = (f0 + j) * hop + n - pad;  // index into the unpadded signal
        float v = 0.f;
        if (f0 + j < frames) {
            if (reflect) {  // numpy 'reflect': the edge sample is not repeated
                if (p < 0) p = -p;
                if (p >= L) p = 2 * (L - 1) - p;
            }
            if (p >= 0 && p < L) v = fminf(fmaxf(w[p], -1.f), 1.f);
        }
        const float hann = 0.5f - 0.5f * tw[n].x;  // periodic Hann
        xw[i] = v * hann;
    }
    __syncthreads();
    const int k = threadIdx.x;
    if (k >= bins) return;
    float re[LS_FPB], im[LS_FPB];
#pragma unroll
    for (int j = 0; j < LS_FPB; ++j) re[j] = 0.f, im[j] = 0.f;
    const int msk = n_fft - 1;
    int idx = 0;  // (k * n) mod n_fft
#pragma unroll 4
    for (int n = 0; n < n_fft; ++n) {
        const float2 t = tw[idx];
#pragma unroll
        for (int j = 0; j < LS_FPB; ++j) {
            const float x = xw[j * n_fft + n];
            re[j] = fmaf(x, t.x, re[j]);
            im[j] = fmaf(-x, t.y, im[j]);
        }
        idx = (idx + k) & msk;
    }
    float* o = out + ((size_t)b * bins + k) * frames + f0;
#pragma unroll
    for (int j = 0; j < LS_FPB; ++j)
        if (f0 + j < frames) o[j] = logf(sqrtf(re[j] * re[j] + im[j] * im[j]) + 1e-7f);
}

int logspec_frames(int L, int hop) { return 1 + L / hop; }

int logspec(const float* wave, int B, int L, int n_fft, int hop, int reflect, float* out, hipStream_t st) {
    const int frames = logspec_frames(L, hop), bins = n_fft / 2 + 1;
    const int threads = (bins + 63) / 64 * 64;
    const size_t lds = (size_t)n_fft * sizeof(float2) + (size_t)LS_FPB * n_fft * sizeof(float);
    hipLaunchKernelGGL(logspec_kernel, dim3(ceil_div(frames, LS_FPB), B), dim3(threads), lds, st, wave, out, L, n_fft, hop, frames,
                       reflect);
    GDL_CHECK_LAUNCH("logspec_kernel");
    return GDL_OK;
}

struct Norm3 {
    float mean[3], std[3];
};
__global__ __launch_bounds__(256) void frames_normalize_kernel(const unsigned char* __restrict__ in, float* __restrict__ out,
                                                              size_t n_img, int hw, Norm3 nm) {
    // one thread per pixel: 3 bytes in, one float into each of the 3 channel planes
    const size_t total = n_img * (size_t)hw;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t n = i / hw, p = i - n * hw;
        const unsigned char* s = in + i * 3;
        float* o = out + n * 3 * (size_t)hw + p;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(size_t)c * hw] = ((float)s[c] / 255.f - nm.mean[c]) / nm.std[c];
    }
}

int frames_normalize(const unsigned char* in, size_t n_img, int H, int W, const float* mean, const float* std, float* out,
                     hipStream_t st) {
    Norm3 nm;
    for (int c = 0; c < 3; ++c) nm.mean[c] = mean[c], nm.std[c] = std[c];
    const size_t total = n_img * (size_t)H * W;
    const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(frames_normalize_kernel, dim3(grid), dim3(256), 0, st, in, out, n_img, H * W, nm);
    GDL_CHECK_LAUNCH("frames_normalize_kernel");
    return GDL_OK;
}

}  // namespace gdl
