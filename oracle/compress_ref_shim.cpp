/*
 * compress_ref_shim.cpp — extern "C" access to the REFERENCE's CompressHelper (compiled from
 * /root/reference/Compression/CompressHelper.cpp by oracle/Makefile into oracle/_ref/libcompress_ref.so).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/golden/make_compress_golden.py (fixture generation, in the build
 * container) and tests/test_compress_oracle.py to pin oracle/kwave_oracle.c's restatement of the basis and to
 * produce golden vectors.  Contains no reference code: it only calls the public API declared at
 * Compression/CompressHelper.h:72-92.
 */
#include <Compression/CompressHelper.h>

#include <cstring>

extern "C" {

void cref_init(float period, unsigned long long mos, unsigned long long harmonics)
{
  // Parameters.cpp:549-551 calls init(..., normalize = true)
  CompressHelper::getInstance().init(period, mos, harmonics, true);
}

unsigned long long cref_osize() { return CompressHelper::getInstance().getOSize(); }
unsigned long long cref_bsize() { return CompressHelper::getInstance().getBSize(); }

/* which: 0 bE, 1 bE_1, 2 bEShifted, 3 bE_1Shifted; out: [harmonics*bSize] complex (interleaved) */
void cref_basis(int which, float* out)
{
  CompressHelper& h = CompressHelper::getInstance();
  const FloatComplex* src = nullptr;
  switch (which)
  {
    case 0: src = h.getBE(); break;
    case 1: src = h.getBE_1(); break;
    case 2: src = h.getBEShifted(); break;
    default: src = h.getBE_1Shifted(); break;
  }
  std::memcpy(out, src, sizeof(FloatComplex) * h.getHarmonics() * h.getBSize());
}

float cref_find_period(const float* data, unsigned long long length)
{
  return CompressHelper::findPeriod(data, length);
}

void cref_to40b(float re, float im, unsigned char* out5, int e)
{
  CompressHelper::convertFloatCTo40b(FloatComplex(re, im), out5, e);
}
void cref_from40b(unsigned char* in5, float* re_im, int e)
{
  FloatComplex c;
  CompressHelper::convert40bToFloatC(in5, c, e);
  re_im[0] = c.real();
  re_im[1] = c.imag();
}
}
