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

// cos/sin(2*pi*k/48), k = 0..47 (compile-time twiddles for DFT sizes 3, 6, 12, 24, 48)
__device__ constexpr float kCos48[48] = {
  1.0f, 0.9914448613738104f, 0.9659258262890683f, 0.9238795325112867f, 0.8660254037844387f, 0.7933533402912352f,
  0.7071067811865476f, 0.6087614290087207f, 0.5000000000000001f, 0.38268343236508984f, 0.25881904510252074f,
  0.1305261922200517f, 0.0f, -0.1305261922200516f, -0.25881904510252063f, -0.3826834323650895f,
  -0.4999999999999998f, -0.6087614290087207f, -0.7071067811865475f, -0.793353340291235f, -0.8660254037844387f,
  -0.9238795325112867f, -0.9659258262890682f, -0.9914448613738104f, -1.0f, -0.9914448613738104f,
  -0.9659258262890683f, -0.9238795325112868f, -0.8660254037844388f, -0.7933533402912352f, -0.7071067811865479f,
  -0.6087614290087209f, -0.5000000000000004f, -0.3826834323650895f, -0.25881904510252063f, -0.13052619222005163f,
  0.0f, 0.13052619222005127f, 0.2588190451025203f, 0.38268343236508917f, 0.5000000000000001f,
  0.6087614290087199f, 0.7071067811865474f, 0.7933533402912349f, 0.8660254037844384f, 0.9238795325112868f,
  0.9659258262890681f, 0.9914448613738104f };
__device__ constexpr float kSin48[48] = {
  0.0f, 0.13052619222005157f, 0.25881904510252074f, 0.3826834323650898f, 0.49999999999999994f, 0.6087614290087207f,
  0.7071067811865475f, 0.7933533402912352f, 0.8660254037844386f, 0.9238795325112867f, 0.9659258262890683f,
  0.9914448613738104f, 1.0f, 0.9914448613738104f, 0.9659258262890683f, 0.9238795325112868f, 0.8660254037844387f,
  0.7933533402912352f, 0.7071067811865476f, 0.6087614290087209f, 0.49999999999999994f, 0.3826834323650899f,
  0.258819045102521f, 0.130526192220052f, 0.0f, -0.13052619222005177f, -0.2588190451025208f,
  -0.38268343236508967f, -0.4999999999999997f, -0.6087614290087207f, -0.7071067811865471f, -0.7933533402912349f,
  -0.8660254037844384f, -0.9238795325112868f, -0.9659258262890683f, -0.9914448613738104f, -1.0f,
  -0.9914448613738105f, -0.9659258262890684f, -0.923879532511287f, -0.8660254037844386f, -0.7933533402912357f,
  -0.7071067811865477f, -0.6087614290087209f, -0.5000000000000004f, -0.38268343236508956f, -0.25881904510252157f,
  -0.13052619222005168f };

// cos/sin(2*pi*k/80), k = 0..79 (compile-time twiddles for DFT sizes 5, 10, 20, 40)
__device__ constexpr float kCos80[80] = {
  1.0f, 0.996917333733128f, 0.9876883405951378f, 0.9723699203976766f, 0.9510565162951535f, 0.9238795325112867f,
  0.8910065241883679f, 0.8526401643540922f, 0.8090169943749475f, 0.7604059656000309f, 0.7071067811865476f,
  0.6494480483301838f, 0.5877852522924731f, 0.5224985647159489f, 0.4539904997395468f, 0.38268343236508984f,
  0.30901699437494745f, 0.23344536385590547f, 0.15643446504023092f, 0.078459095727845f, 0.0f,
  -0.07845909572784487f, -0.1564344650402306f, -0.23344536385590534f, -0.30901699437494734f, -0.3826834323650897f,
  -0.4539904997395467f, -0.5224985647159488f, -0.587785252292473f, -0.6494480483301835f, -0.7071067811865475f,
  -0.7604059656000309f, -0.8090169943749473f, -0.8526401643540922f, -0.8910065241883678f, -0.9238795325112867f,
  -0.9510565162951535f, -0.9723699203976766f, -0.9876883405951377f, -0.996917333733128f, -1.0f,
  -0.9969173337331281f, -0.9876883405951378f, -0.9723699203976767f, -0.9510565162951538f, -0.9238795325112868f,
  -0.8910065241883679f, -0.8526401643540921f, -0.8090169943749476f, -0.7604059656000314f, -0.7071067811865477f,
  -0.6494480483301841f, -0.5877852522924732f, -0.5224985647159486f, -0.4539904997395469f, -0.3826834323650895f,
  -0.30901699437494756f, -0.233445363855906f, -0.15643446504023104f, -0.07845909572784557f, 0.0f,
  0.07845909572784521f, 0.15643446504023067f, 0.23344536385590567f, 0.30901699437494723f, 0.38268343236508917f,
  0.45399049973954664f, 0.5224985647159484f, 0.5877852522924729f, 0.6494480483301839f, 0.7071067811865474f,
  0.760405965600031f, 0.8090169943749473f, 0.8526401643540918f, 0.8910065241883678f, 0.9238795325112865f,
  0.9510565162951535f, 0.9723699203976767f, 0.9876883405951377f, 0.996917333733128f };
__device__ constexpr float kSin80[80] = {
  0.0f, 0.07845909572784494f, 0.15643446504023087f, 0.2334453638559054f, 0.3090169943749474f, 0.3826834323650898f,
  0.45399049973954675f, 0.5224985647159488f, 0.5877852522924731f, 0.6494480483301837f, 0.7071067811865475f,
  0.7604059656000308f, 0.8090169943749475f, 0.8526401643540922f, 0.8910065241883678f, 0.9238795325112867f,
  0.9510565162951535f, 0.9723699203976766f, 0.9876883405951378f, 0.996917333733128f, 1.0f, 0.996917333733128f,
  0.9876883405951378f, 0.9723699203976767f, 0.9510565162951536f, 0.9238795325112867f, 0.8910065241883679f,
  0.8526401643540923f, 0.8090169943749475f, 0.760405965600031f, 0.7071067811865476f, 0.6494480483301838f,
  0.5877852522924732f, 0.5224985647159489f, 0.45399049973954686f, 0.3826834323650899f, 0.3090169943749475f,
  0.23344536385590553f, 0.15643446504023098f, 0.07845909572784507f, 0.0f, -0.07845909572784437f,
  -0.15643446504023073f, -0.23344536385590528f, -0.3090169943749469f, -0.38268343236508967f, -0.4539904997395467f,
  -0.5224985647159491f, -0.587785252292473f, -0.6494480483301832f, -0.7071067811865475f, -0.7604059656000306f,
  -0.8090169943749473f, -0.8526401643540924f, -0.8910065241883678f, -0.9238795325112868f, -0.9510565162951535f,
  -0.9723699203976764f, -0.9876883405951377f, -0.996917333733128f, -1.0f, -0.996917333733128f,
  -0.9876883405951378f, -0.9723699203976766f, -0.9510565162951536f, -0.923879532511287f, -0.891006524188368f,
  -0.8526401643540925f, -0.8090169943749476f, -0.7604059656000308f, -0.7071067811865477f, -0.6494480483301834f,
  -0.5877852522924734f, -0.5224985647159495f, -0.45399049973954697f, -0.3826834323650904f, -0.3090169943749476f,
  -0.2334453638559052f, -0.15643446504023112f, -0.07845909572784475f };

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
  constexpr int  TN  = (R % 3 == 0) ? 48 : (R % 5 == 0) ? 80 : 64;  // table the angle k/R lives in
  constexpr int  idx = (K * (TN / R)) % TN;
  static_assert(TN % R == 0, "DFT size must divide 64, 48 or 80");
  if (idx == 0) return a;
  if (2 * idx == TN) return make_float2(-a.x, -a.y);
  if (4 * idx == TN) return (DIR < 0) ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);   // * (-/+ i)
  if (4 * idx == 3 * TN) return (DIR < 0) ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
  const float c = (TN == 48) ? kCos48[idx] : (TN == 80) ? kCos80[idx % 80] : kCos64[idx % 64];
  const float t = (TN == 48) ? kSin48[idx] : (TN == 80) ? kSin80[idx % 80] : kSin64[idx % 64];
  const float s = (DIR < 0) ? -t : t;
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
  // sizes 3 * 2^m: one radix-3 decimation-in-time stage over three power-of-two sub-transforms
  template<int K> static __device__ __forceinline__ void combine3(float2 (&v)[R], const float2 (&e0)[R / 3],
                                                                 const float2 (&e1)[R / 3], const float2 (&e2)[R / 3])
  {
    if constexpr (K < R / 3)
    {
      constexpr int M = R / 3;
      const float2 t1 = mul_const_tw<R, K, DIR>(e1[K]);
      const float2 t2 = mul_const_tw<R, (2 * K) % R, DIR>(e2[K]);
      const float2 s  = cadd(t1, t2);
      const float2 d  = csub(t1, t2);
      const float2 m  = make_float2(e0[K].x - 0.5f * s.x, e0[K].y - 0.5f * s.y);
      // DIR * i * (sqrt(3)/2) * d
      constexpr float h = 0.86602540378443864676f;
      const float2 n  = (DIR < 0) ? make_float2(h * d.y, -h * d.x) : make_float2(-h * d.y, h * d.x);
      v[K]         = cadd(e0[K], s);
      v[K + M]     = cadd(m, n);
      v[K + 2 * M] = csub(m, n);
      combine3<K + 1>(v, e0, e1, e2);
    }
  }
  // sizes 5 * 2^m: one radix-5 decimation-in-time stage over five power-of-two sub-transforms
  template<int K> static __device__ __forceinline__ void combine5(float2 (&v)[R], const float2 (&e)[5][R / 5])
  {
    if constexpr (K < R / 5)
    {
      constexpr int   M  = R / 5;
      constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;   // cos(2 pi/5), cos(4 pi/5)
      constexpr float n1 = 0.95105651629515357212f, n2 = 0.58778525229247312917f;    // sin(2 pi/5), sin(4 pi/5)
      const float2 a0 = e[0][K];
      const float2 a1 = mul_const_tw<R, K, DIR>(e[1][K]);
      const float2 a2 = mul_const_tw<R, (2 * K) % R, DIR>(e[2][K]);
      const float2 a3 = mul_const_tw<R, (3 * K) % R, DIR>(e[3][K]);
      const float2 a4 = mul_const_tw<R, (4 * K) % R, DIR>(e[4][K]);
      const float2 s1 = cadd(a1, a4), s2 = cadd(a2, a3), d1 = csub(a1, a4), d2 = csub(a2, a3);
      const float2 m1 = make_float2(a0.x + c1 * s1.x + c2 * s2.x, a0.y + c1 * s1.y + c2 * s2.y);
      const float2 m2 = make_float2(a0.x + c2 * s1.x + c1 * s2.x, a0.y + c2 * s1.y + c1 * s2.y);
      const float2 q1 = make_float2(n1 * d1.x + n2 * d2.x, n1 * d1.y + n2 * d2.y);
      const float2 q2 = make_float2(n2 * d1.x - n1 * d2.x, n2 * d1.y - n1 * d2.y);
      // forward: X1 = m1 - i q1, X4 = m1 + i q1, X2 = m2 - i q2, X3 = m2 + i q2 (inverse: conjugate signs)
      const float2 iq1 = (DIR < 0) ? make_float2(q1.y, -q1.x) : make_float2(-q1.y, q1.x);
      const float2 iq2 = (DIR < 0) ? make_float2(q2.y, -q2.x) : make_float2(-q2.y, q2.x);
      v[K]         = make_float2(a0.x + s1.x + s2.x, a0.y + s1.y + s2.y);
      v[K + M]     = cadd(m1, iq1);
      v[K + 2 * M] = cadd(m2, iq2);
      v[K + 3 * M] = csub(m2, iq2);
      v[K + 4 * M] = csub(m1, iq1);
      combine5<K + 1>(v, e);
    }
  }
  static __device__ __forceinline__ void run(float2 (&v)[R])
  {
#ifdef KW_NO_DFT /* timing experiment only: how much of a pass is butterfly arithmetic (results are wrong) */
    return;
#endif
    if constexpr (R % 3 == 0)
    {
      float2 e0[R / 3], e1[R / 3], e2[R / 3];
#pragma unroll
      for (int k = 0; k < R / 3; k++) { e0[k] = v[3 * k]; e1[k] = v[3 * k + 1]; e2[k] = v[3 * k + 2]; }
      Dft<R / 3, DIR>::run(e0);
      Dft<R / 3, DIR>::run(e1);
      Dft<R / 3, DIR>::run(e2);
      combine3<0>(v, e0, e1, e2);
      return;
    }
    if constexpr (R % 5 == 0)
    {
      float2 e5[5][R / 5];
#pragma unroll
      for (int k = 0; k < R / 5; k++)
      {
#pragma unroll
        for (int r = 0; r < 5; r++) e5[r][k] = v[5 * k + r];
      }
#pragma unroll
      for (int r = 0; r < 5; r++) Dft<R / 5, DIR>::run(e5[r]);
      combine5<0>(v, e5);
      return;
    }
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
template<> struct Fac<48>   { static constexpr int R1 = 6,  R2 = 8; };
template<> struct Fac<72>   { static constexpr int R1 = 6,  R2 = 12; };
template<> struct Fac<80>   { static constexpr int R1 = 8,  R2 = 10; };
template<> struct Fac<100>  { static constexpr int R1 = 10, R2 = 10; };
template<> struct Fac<120>  { static constexpr int R1 = 10, R2 = 12; };
template<> struct Fac<200>  { static constexpr int R1 = 10, R2 = 20; };
template<> struct Fac<400>  { static constexpr int R1 = 20, R2 = 20; };
template<> struct Fac<144>  { static constexpr int R1 = 12, R2 = 12; };
template<> struct Fac<160>  { static constexpr int R1 = 10, R2 = 16; };
template<> struct Fac<240>  { static constexpr int R1 = 12, R2 = 20; };
template<> struct Fac<288>  { static constexpr int R1 = 12, R2 = 24; };
template<> struct Fac<320>  { static constexpr int R1 = 16, R2 = 20; };
template<> struct Fac<480>  { static constexpr int R1 = 20, R2 = 24; };
template<> struct Fac<576>  { static constexpr int R1 = 24, R2 = 24; };
template<> struct Fac<640>  { static constexpr int R1 = 20, R2 = 32; };
template<> struct Fac<768>  { static constexpr int R1 = 24, R2 = 32; };
template<> struct Fac<96>   { static constexpr int R1 = 8,  R2 = 12; };
template<> struct Fac<192>  { static constexpr int R1 = 12, R2 = 16; };
template<> struct Fac<384>  { static constexpr int R1 = 16, R2 = 24; };
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
