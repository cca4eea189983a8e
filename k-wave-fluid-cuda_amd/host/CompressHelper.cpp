// CompressHelper.cpp — see CompressHelper.h (restates Compression/CompressHelper.cpp:48-65,672-778).
#include "CompressHelper.h"

#include <cmath>

CompressHelper& CompressHelper::getInstance()
{
  static CompressHelper instance;
  return instance;
}

void CompressHelper::init(float period, size_t mos, size_t harmonics, bool normalize)
{
  mOSize     = size_t(period * mos);
  mBSize     = mOSize * 2 + 1;
  mPeriod    = period;
  mMos       = mos;
  mHarmonics = harmonics;
  generateFunctions(mBE, mBE_1, normalize, false);
  generateFunctions(mBEShifted, mBE_1Shifted, normalize, true);
}

void CompressHelper::generateFunctions(std::vector<FloatComplex>& bE, std::vector<FloatComplex>& bE_1, bool normalize,
                                       bool shift) const
{
  const size_t bSize = mBSize, oSize = mOSize;
  std::vector<float> b(bSize);
  // triangular window (:700-710)
  for (size_t x = 0; x < oSize; x++) b[x] = float(x) / oSize;
  for (size_t x = oSize; x < 2 * oSize + 1; x++) b[x] = 2.0f - float(x) / oSize;
  std::vector<FloatComplex> e(mHarmonics * bSize);
  bE.assign(mHarmonics * bSize, FloatComplex());
  bE_1.assign(mHarmonics * bSize, FloatComplex());
  const FloatComplex i(0.0f, -1.0f);
  for (size_t ih = 0; ih < mHarmonics; ih++)
  {
    const size_t h = ih + 1;
    for (size_t x = 0; x < bSize; x++)
    { // generateE (:733-746)
      const size_t hx = ih * bSize + x;
      e[hx]           = std::exp(i * (2.0f * float(M_PI) / (mPeriod / float(h))) * float(x));
      if (shift) e[hx] *= std::exp(-i * float(M_PI) / (mPeriod / float(h)));
    }
    for (size_t x = 0; x < bSize; x++)
    { // generateBE (:760-778)
      const size_t hx = ih * bSize + x;
      bE[hx]          = b[x] * e[hx];
      bE_1[hx]        = b[(x + oSize) % (bSize - 1)] * e[ih * bSize + ((x + oSize) % (bSize - 1))];
      if (normalize)
      {
        bE[hx] *= (2.0f / float(oSize));
        bE_1[hx] *= (2.0f / float(oSize));
      }
    }
  }
}
