// CompressHelper.h — compression basis for the on-the-fly time-series compression streams.
// Mirror of Compression/CompressHelper.{h,cpp} of the reference for what the sampling path needs: init (:48-65),
// the triangular window (:700-710), the complex exponential basis (:733-746, velocity streams phase-shifted by half a
// step) and the windowed / inverted-window bases bE, bE_1 (:760-778), normalised by 2/oSize as Parameters.cpp:549-551
// requests.  The period finder and the 40-bit codec (:146-389) are later scope rows.
#ifndef KW_HOST_COMPRESS_HELPER_H
#define KW_HOST_COMPRESS_HELPER_H
#include <complex>
#include <cstddef>
#include <vector>

using FloatComplex = std::complex<float>;

class CompressHelper
{
 public:
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

 private:
  CompressHelper() = default;
  void generateFunctions(std::vector<FloatComplex>& bE, std::vector<FloatComplex>& bE_1, bool normalize, bool shift) const;
  size_t mOSize = 0, mBSize = 0, mMos = 1, mHarmonics = 1;
  float  mPeriod = 0.0f;
  std::vector<FloatComplex> mBE, mBEShifted, mBE_1, mBE_1Shifted;
};
#endif
