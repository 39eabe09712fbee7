// Audio front-end ("next" row f-2): the steps immediately upstream of AudioMAE,
// reference audio_encoder/AudioMAE.py:356-394 (extract_kaldi_fbank_feature):
//   apad_resample_fir   torchaudio.functional.resample as a polyphase FIR (kernel table from the host)
//   apad_kaldi_fbank    Kaldi-compatible 128-bin log-mel filterbank of 25 ms / 10 ms frames, zero-padded or cropped to
//                       `target_frames` rows BEFORE the (x - mean) / (2 std) normalisation, like the reference
// fp32 throughout (the reference computes the mel in fp32 and feeds AudioMAE fp32).  HBM/latency-bound byte-sized work:
// one workgroup per frame, the whole frame lives in LDS (DC removal, pre-emphasis, window, 512-point radix-2 FFT, power,
// mel projection, log); nothing here is shaped for MFMA.
#include "common.h"

namespace {

constexpr int WIN = 400, SHIFT = 160, NFFT = 512, NBIN = 257;

__global__ __launch_bounds__(256) void resample_kernel(const float* x, const float* kern, float* out, int64_t n_in, int64_t n_out,
                                                       int orig, int newf, int width, int kw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const int64_t blk = i / newf;
    const int phase = (int)(i - blk * newf);
    const float* kp = kern + (int64_t)phase * kw;
    const int64_t base = blk * orig - width;  // index into the un-padded input
    float acc = 0.f;
    for (int j = 0; j < kw; ++j) {
        const int64_t s = base + j;
        const float v = (s >= 0 && s < n_in) ? x[s] : 0.f;
        acc = fmaf(kp[j], v, acc);
    }
    out[i] = acc;
}

// sum over the workgroup in a fixed order
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void fbank_kernel(const float* x, float dc, const float* window, const float* twiddle /* [256][2] */,
                                                    const float* mel /* [nmel][257] */, float* out, int n_frames, int nmel,
                                                    float preemph, float norm_mean, float inv_2std) {
    __shared__ float re[NFFT], im[NFFT], raw[WIN], sh[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    if (f >= n_frames) {  // rows past the last frame: zero-padded BEFORE normalisation (AudioMAE.py:384-387, :393)
        for (int b = tid; b < nmel; b += 256) out[(int64_t)f * nmel + b] = (0.f - norm_mean) * inv_2std;
        return;
    }
    const float* xf = x + (int64_t)f * SHIFT;
    float part = 0.f;
    for (int j = tid; j < WIN; j += 256) {
        const float v = xf[j] - dc;  // waveform - waveform.mean() (:368)
        raw[j] = v;
        part += v;
    }
    const float mean = block_sum(part, sh) / (float)WIN;  // remove_dc_offset, per frame
    // bit-reversed load for the in-place decimation-in-time FFT; samples >= WIN are the zero padding to 512
    for (int j = tid; j < NFFT; j += 256) {
        float v = 0.f;
        if (j < WIN) {
            const float cur = raw[j] - mean, prev = raw[j > 0 ? j - 1 : 0] - mean;  // replicate-padded pre-emphasis
            v = (cur - preemph * prev) * window[j];
        }
        const int r = (int)(__brev((unsigned)j) >> 23);  // 9-bit reversal
        re[r] = v;
        im[r] = 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 1; s <= 9; ++s) {
        const int half = 1 << (s - 1);
        const int grp = tid >> (s - 1), pos = tid & (half - 1);
        const int i0 = grp * (half << 1) + pos, i1 = i0 + half;
        const int tw = pos << (9 - s);  // twiddle index k * (512 / 2^s)
        const float wr = twiddle[2 * tw], wi = twiddle[2 * tw + 1];
        const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
        const float tr = br * wr - bi * wi, ti = br * wi + bi * wr;
        re[i0] = ar + tr; im[i0] = ai + ti;
        re[i1] = ar - tr; im[i1] = ai - ti;
        __syncthreads();
    }
    // power spectrum, bins 0..256 (reuse re[] for the power; bin 256 lives in re[256])
    float p0 = re[tid] * re[tid] + im[tid] * im[tid];
    float p256 = 0.f;
    if (tid == 0) p256 = re[256] * re[256] + im[256] * im[256];
    __syncthreads();
    re[tid] = p0;
    if (tid == 0) re[256] = p256;
    __syncthreads();
    for (int b = tid; b < nmel; b += 256) {
        const float* w = mel + (int64_t)b * NBIN;
        float e = 0.f;
        for (int k = 0; k < NBIN; ++k) e = fmaf(w[k], re[k], e);
        e = fmaxf(e, 1.1920928955078125e-07f);  // use_log_fbank: max(eps).log()
        out[(int64_t)f * nmel + b] = (__logf(e) - norm_mean) * inv_2std;
    }
}

}  // namespace

extern "C" int apad_resample_fir(const float* x, const float* kernel, float* out, int64_t n_in, int64_t n_out, int32_t orig,
                                 int32_t newf, int32_t width, void* stream) {
    APAD_CHECK(x && kernel && out && n_in > 0 && n_out > 0 && orig > 0 && newf > 0 && width >= 0, "apad_resample_fir: bad operands");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, x, kernel, out, n_in, n_out, orig, newf,
                       width, 2 * width + orig);
    return apad_check_launch("apad_resample_fir");
}

extern "C" int apad_kaldi_fbank(const float* x, int64_t n_samples, float dc, const float* window, const float* twiddle, const float* mel,
                                float* out, int32_t target_frames, int32_t num_mel_bins, float preemphasis, float norm_mean,
                                float norm_std, void* stream) {
    APAD_CHECK(x && window && twiddle && mel && out && target_frames > 0 && num_mel_bins > 0, "apad_kaldi_fbank: bad operands");
    int64_t frames = n_samples < WIN ? 0 : 1 + (n_samples - WIN) / SHIFT;  // snip_edges
    if (frames > target_frames) frames = target_frames;                     // crop (:388-389)
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)target_frames), dim3(256), 0, s, x, dc, window, twiddle, mel, out, (int)frames,
                       num_mel_bins, preemphasis, norm_mean, 1.0f / (2.0f * norm_std));
    return apad_check_launch("apad_kaldi_fbank");
}
