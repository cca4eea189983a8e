// CompressHelper.cpp — see CompressHelper.h (restates Compression/CompressHelper.cpp:48-65,672-778).
#include "CompressHelper.h"
#include "Parameters.h"
#include <stdexcept>
#include <limits>
#include <cstring>
#include <algorithm>

#include <cmath>

CompressHelper& CompressHelper::getInstance() { return Parameters::getInstance().getCompressHelper(); }

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

// ---- period finder ----------------------------------------------------------------------------------------------------
float CompressHelper::findPeriod(const float* x, size_t n)
{
  // local maxima with a parabolic-free, slope-ratio sub-sample position (findPeaks, :549-572)
  std::vector<float> locs, peaks;
  for (size_t i = 1; i + 1 < n; i++)
  {
    if (x[i] > x[i - 1] && x[i] >= x[i + 1])
    {
      const float d1 = x[i] - x[i - 1];
      const float d2 = x[i] - x[i + 1];
      locs.push_back(float(i) + d1 / (d1 + d2) - 0.5f);
      peaks.push_back(x[i]);
    }
  }
  // keep the peaks above half of the largest one (:160-179; the search starts from FLT_MIN like the reference)
  float top = std::numeric_limits<float>::min();
  for (float v : peaks) top = std::max(top, v);
  std::vector<float> kept;
  for (size_t i = 0; i < peaks.size(); i++)
    if (peaks[i] > 0.5f * top) kept.push_back(locs[i]);
  if (kept.size() < 2) throw std::invalid_argument("findPeriod: fewer than two peaks in the signal");
  // median of the spacings: element length/2 of the sorted differences (:181-183, :640-645)
  std::vector<float> gaps(kept.size() - 1);
  for (size_t i = 0; i + 1 < kept.size(); i++) gaps[i] = kept[i + 1] - kept[i];
  std::sort(gaps.begin(), gaps.end());
  return gaps[gaps.size() / 2];
}

// ---- 40-bit codec -----------------------------------------------------------------------------------------------------
namespace
{
inline uint32_t bitsOf(float v) { uint32_t u; std::memcpy(&u, &v, 4); return u; }
inline float    floatOf(uint32_t u) { float v; std::memcpy(&v, &u, 4); return v; }

// one component of the encoder: 23-bit fraction -> 17-bit field (explicit leading bit), shifted right by `shift`
inline uint32_t packMantissa(uint32_t fraction, uint32_t shift)
{
  uint32_t m = fraction >> shift;
  if (m > 0 && m != (0x7FFFFFu >> shift)) m++; // round up unless that would carry out of the field (:351-365)
  m |= 1u << (23 - shift);                     // leading one of the float becomes an explicit flag bit
  return m >> 1;                               // 17 bits
}

// one component of the decoder: 17-bit field + shared exponent -> IEEE-754 bits
inline uint32_t unpackComponent(uint32_t field17, uint32_t sign, int32_t exponent)
{
  uint32_t m = field17 << 6; // back to 23 bits
  if (m == 0) return sign << 31; // exponent 0, fraction 0
  const int index = 31 - __builtin_clz(m); // position of the flag bit
  m <<= 23 - index;
  exponent -= 22 - index;
  return (sign << 31) | (uint32_t(exponent) << 23) | (m & 0x007FFFFFu);
}
} // namespace

void CompressHelper::convertFloatCTo40b(FloatComplex value, uint8_t* out, int32_t e)
{
  const uint32_t bR = bitsOf(value.real()), bI = bitsOf(value.imag());
  const uint32_t sR = bR >> 31, sI = bI >> 31;
  const int32_t  eR = int32_t((bR & 0x7F800000u) >> 23) - e, eI = int32_t((bI & 0x7F800000u) >> 23) - e;
  // shared exponent = the larger one; the smaller component is shifted right by the difference; 6 bits are dropped
  int32_t  eS = std::max(eR, eI);
  uint32_t shiftR = 6 + uint32_t(eS - eR), shiftI = 6 + uint32_t(eS - eI);
  if (eS < 0)
  { // below the representable range: denormalise against exponent 0 (:337-344)
    shiftR += uint32_t(-eS);
    shiftI += uint32_t(-eS);
    eS = 0;
  }
  shiftR = std::min(shiftR, 23u);
  shiftI = std::min(shiftI, 23u);
  uint32_t mR = packMantissa(bR & 0x007FFFFFu, shiftR);
  uint32_t mI = packMantissa(bI & 0x007FFFFFu, shiftI);
  if (eS > 0xF)
  { // above the range: saturate (:376-381)
    mR = mI = 0xFFFF;
    eS = 0xF;
  }
  out[0] = uint8_t((sR << 7) | (sI << 6) | ((mR & 0x10000u) >> 11) | ((mI & 0x10000u) >> 12) | (uint32_t(eS) & 0xFu));
  out[1] = uint8_t(mR & 0xFFu);
  out[2] = uint8_t((mR >> 8) & 0xFFu);
  out[3] = uint8_t(mI & 0xFFu);
  out[4] = uint8_t((mI >> 8) & 0xFFu);
}

void CompressHelper::convert40bToFloatC(const uint8_t* in, FloatComplex& value, int32_t e)
{
  const uint32_t head = in[0];
  const uint32_t mR = ((head & 0x20u) << 11) | (uint32_t(in[2]) << 8) | in[1];
  const uint32_t mI = ((head & 0x10u) << 12) | (uint32_t(in[4]) << 8) | in[3];
  const int32_t  ex = int32_t(head & 0xFu) + e;
  value = FloatComplex(floatOf(unpackComponent(mR, head >> 7, ex)), floatOf(unpackComponent(mI, (head & 0x40u) >> 6, ex)));
}
