// CompressHelper.h — compression basis for the on-the-fly time-series compression streams.
// Mirror of Compression/CompressHelper.{h,cpp} of the reference for what the sampling path needs: init (:48-65),
// the triangular window (:700-710), the complex exponential basis (:733-746, velocity streams phase-shifted by half a
// step) and the windowed / inverted-window bases bE, bE_1 (:760-778), normalised by 2/oSize as Parameters.cpp:549-551
// requests; the period finder (:146-216, with findPeaks :549-572, diff, median) used when no --period is given
// (Parameters.cpp:488-512) and the 40-bit packing of complex coefficients (:224-389).
#ifndef KW_HOST_COMPRESS_HELPER_H
#define KW_HOST_COMPRESS_HELPER_H
#include <complex>
#include <cstddef>
#include <cstdint>
#include <vector>

using FloatComplex = std::complex<float>;

class CompressHelper
{
 public:
  /// the basis of the parameter set the calling thread works on (Parameters::getInstance)
  static CompressHelper& getInstance();
  void init(float period, size_t mos, size_t harmonics, bool normalize = false);
  const FloatComplex* getBE() const { return mBE.data(); }
  const FloatComplex* getBEShifted() const { return mBEShifted.data(); }
  const FloatComplex* getBE_1() const { return mBE_1.data(); }
  const FloatComplex* getBE_1Shifted() const { return mBE_1Shifted.data(); }
  size_t getOSize() const { return mOSize; }
  size_t getBSize() const { return mBSize; }
  float  getPeriod() const { return mPeriod; }
  size_t getMos() const { return mMos; }
  size_t getHarmonics() const { return mHarmonics; }

  /// period of a sampled signal in samples: sub-sample positions of the local maxima above half the largest one,
  /// median of their spacings (CompressHelper.cpp:146-216)
  static float findPeriod(const float* data, size_t length);
  /// 40-bit packing of a complex coefficient (CompressHelper.cpp:298-389): byte 0 = real sign | imaginary sign |
  /// bit 16 of each mantissa | 4-bit shared exponent (biased by `e`: 138 for pressure, 114 for velocity),
  /// bytes 1-2 / 3-4 = low 16 bits of the real / imaginary 17-bit mantissa (leading flag bit explicit)
  static void convertFloatCTo40b(FloatComplex value, uint8_t* packed5, int32_t e);
  static void convert40bToFloatC(const uint8_t* packed5, FloatComplex& value, int32_t e);
  static constexpr int32_t kMaxExpP = 138, kMaxExpU = 114; // CompressHelper.h:81-83

 private:
  friend class Parameters; // one helper per parameter set
  CompressHelper() = default;
  void generateFunctions(std::vector<FloatComplex>& bE, std::vector<FloatComplex>& bE_1, bool normalize, bool shift) const;
  size_t mOSize = 0, mBSize = 0, mMos = 1, mHarmonics = 1;
  float  mPeriod = 0.0f;
  std::vector<FloatComplex> mBE, mBEShifted, mBE_1, mBE_1Shifted;
};
#endif
