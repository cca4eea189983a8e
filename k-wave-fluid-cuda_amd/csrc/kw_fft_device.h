// kw_fft_device.h — in-register / LDS building blocks of the hand-written fused spectral pipeline (gfx950).
//
// A line FFT of length L = R1*R2 is done as two register-resident small DFTs (sizes R1, R2) with ONE exchange
// through LDS ("four-step"):
//   step A  thread (line c, n2):  v[n1] = x[n1*R2 + n2], n1 < R1;  DFT_R1;  v[k1] *= W_L^(n2*k1);  LDS[k1][n2] = v[k1]
//   step B  thread (line c, k1):  w[n2] = LDS[k1][n2],  n2 < R2;  DFT_R2;  w[k2] = X[k1 + R1*k2]
// The inverse started from the step-B register layout needs no extra exchange (roles of R1/R2 swap), which is what
// lets the z-pass do forward FFT -> spectral multiply -> inverse FFT in one kernel.
// Small DFTs are fully unrolled radix-2 decimation-in-time recursions over compile-time twiddles.
#ifndef KW_FFT_DEVICE_H
#define KW_FFT_DEVICE_H

#include <hip/hip_runtime.h>

namespace kwfft {

constexpr int kFwd = -1;
constexpr int kInv = +1;

// cos/sin(2*pi*k/64), k = 0..63 (compile-time twiddles for DFT sizes up to 64)
__device__ constexpr float kCos64[64] = {
  1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f, 0.92387953251128675613f,
  0.88192126434835502971f, 0.83146961230254523708f, 0.77301045336273696081f, 0.70710678118654752440f,
  0.63439328416364549822f, 0.55557023301960222474f, 0.47139673682599764856f, 0.38268343236508977173f,
  0.29028467725446236764f, 0.19509032201612826785f, 0.09801714032956060199f, 0.0f, -0.09801714032956060199f,
  -0.19509032201612826785f, -0.29028467725446236764f, -0.38268343236508977173f, -0.47139673682599764856f,
  -0.55557023301960222474f, -0.63439328416364549822f, -0.70710678118654752440f, -0.77301045336273696081f,
  -0.83146961230254523708f, -0.88192126434835502971f, -0.92387953251128675613f, -0.95694033573220886494f,
  -0.98078528040323044913f, -0.99518472667219688624f, -1.0f, -0.99518472667219688624f, -0.98078528040323044913f,
  -0.95694033573220886494f, -0.92387953251128675613f, -0.88192126434835502971f, -0.83146961230254523708f,
  -0.77301045336273696081f, -0.70710678118654752440f, -0.63439328416364549822f, -0.55557023301960222474f,
  -0.47139673682599764856f, -0.38268343236508977173f, -0.29028467725446236764f, -0.19509032201612826785f,
  -0.09801714032956060199f, 0.0f, 0.09801714032956060199f, 0.19509032201612826785f, 0.29028467725446236764f,
  0.38268343236508977173f, 0.47139673682599764856f, 0.55557023301960222474f, 0.63439328416364549822f,
  0.70710678118654752440f, 0.77301045336273696081f, 0.83146961230254523708f, 0.88192126434835502971f,
  0.92387953251128675613f, 0.95694033573220886494f, 0.98078528040323044913f, 0.99518472667219688624f };
__device__ constexpr float kSin64[64] = {
  0.0f, 0.09801714032956060199f, 0.19509032201612826785f, 0.29028467725446236764f, 0.38268343236508977173f,
  0.47139673682599764856f, 0.55557023301960222474f, 0.63439328416364549822f, 0.70710678118654752440f,
  0.77301045336273696081f, 0.83146961230254523708f, 0.88192126434835502971f, 0.92387953251128675613f,
  0.95694033573220886494f, 0.98078528040323044913f, 0.99518472667219688624f, 1.0f, 0.99518472667219688624f,
  0.98078528040323044913f, 0.95694033573220886494f, 0.92387953251128675613f, 0.88192126434835502971f,
  0.83146961230254523708f, 0.77301045336273696081f, 0.70710678118654752440f, 0.63439328416364549822f,
  0.55557023301960222474f, 0.47139673682599764856f, 0.38268343236508977173f, 0.29028467725446236764f,
  0.19509032201612826785f, 0.09801714032956060199f, 0.0f, -0.09801714032956060199f, -0.19509032201612826785f,
  -0.29028467725446236764f, -0.38268343236508977173f, -0.47139673682599764856f, -0.55557023301960222474f,
  -0.63439328416364549822f, -0.70710678118654752440f, -0.77301045336273696081f, -0.83146961230254523708f,
  -0.88192126434835502971f, -0.92387953251128675613f, -0.95694033573220886494f, -0.98078528040323044913f,
  -0.99518472667219688624f, -1.0f, -0.99518472667219688624f, -0.98078528040323044913f, -0.95694033573220886494f,
  -0.92387953251128675613f, -0.88192126434835502971f, -0.83146961230254523708f, -0.77301045336273696081f,
  -0.70710678118654752440f, -0.63439328416364549822f, -0.55557023301960222474f, -0.47139673682599764856f,
  -0.38268343236508977173f, -0.29028467725446236764f, -0.19509032201612826785f, -0.09801714032956060199f };

#ifdef KW_PK
// complex values as 64-bit register pairs: one v_pk_add_f32 per complex add
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return a + b; }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return a - b; }
#else
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
#endif
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// a * exp(DIR * 2*pi*i * k / R)   (DIR = -1 forward, +1 inverse), compile-time k and R
template<int R, int K, int DIR> __device__ __forceinline__ float2 mul_const_tw(float2 a)
{
  constexpr int idx = (K * (64 / R)) % 64;
  if (idx == 0) return a;
  if (idx == 32) return make_float2(-a.x, -a.y);
  if (idx == 16) return (DIR < 0) ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);   // * (-/+ i) -> exp(-+ i pi/2)
  if (idx == 48) return (DIR < 0) ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
  const float c = kCos64[idx];
  const float s = (DIR < 0) ? -kSin64[idx] : kSin64[idx];
#if defined(KW_PK) && KW_PK >= 2
  return a * make_float2(c, c) + make_float2(-a.y, a.x) * make_float2(s, s);
#else
  return make_float2(a.x * c - a.y * s, a.x * s + a.y * c);
#endif
}

// natural-order in, natural-order out DFT of compile-time size R (power of two <= 64) on registers
template<int R, int DIR> struct Dft
{
  template<int K> static __device__ __forceinline__ void combine(float2 (&v)[R], const float2 (&e)[R / 2], const float2 (&o)[R / 2])
  {
    if constexpr (K < R / 2)
    {
      const float2 t = mul_const_tw<R, K, DIR>(o[K]);
      v[K]           = cadd(e[K], t);
      v[K + R / 2]   = csub(e[K], t);
      combine<K + 1>(v, e, o);
    }
  }
  static __device__ __forceinline__ void run(float2 (&v)[R])
  {
#ifdef KW_NO_DFT /* timing experiment only: how much of a pass is butterfly arithmetic (results are wrong) */
    return;
#endif
    float2 e[R / 2], o[R / 2];
#pragma unroll
    for (int k = 0; k < R / 2; k++) { e[k] = v[2 * k]; o[k] = v[2 * k + 1]; }
    Dft<R / 2, DIR>::run(e);
    Dft<R / 2, DIR>::run(o);
    combine<0>(v, e, o);
  }
};
template<int DIR> struct Dft<1, DIR> { static __device__ __forceinline__ void run(float2 (&)[1]) {} };
template<int DIR> struct Dft<2, DIR>
{
  static __device__ __forceinline__ void run(float2 (&v)[2])
  {
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  }
};

// factorisation L = R1 * R2 used by the pipeline
template<int L> struct Fac;
template<> struct Fac<16>   { static constexpr int R1 = 4,  R2 = 4; };
template<> struct Fac<32>   { static constexpr int R1 = 4,  R2 = 8; };
template<> struct Fac<64>   { static constexpr int R1 = 8,  R2 = 8; };
template<> struct Fac<128>  { static constexpr int R1 = 8,  R2 = 16; };
template<> struct Fac<256>  { static constexpr int R1 = 16, R2 = 16; };
template<> struct Fac<512>  { static constexpr int R1 = 16, R2 = 32; };
template<> struct Fac<1024> { static constexpr int R1 = 32, R2 = 32; };

// inter-step twiddle: table tw[m] = exp(-2*pi*i*m/L) (forward); inverse uses the conjugate
template<int DIR> __device__ __forceinline__ float2 apply_tw(float2 a, float2 w)
{
  if (DIR > 0) w.y = -w.y;
  return cmulf(a, w);
}

} // namespace kwfft
#endif
