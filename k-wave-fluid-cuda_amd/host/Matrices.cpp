// Matrices.cpp — see Matrices.h.
#include "Matrices.h"

#include <cstdlib>
#include <ios>
#include <cstring>

#include "HipError.h"
#include "Parameters.h"

static kw_ctx* ctx() { return Parameters::getInstance().getHipParameters().getContext(); }

// ---- BaseFloatMatrix ------------------------------------------------------------------------------------------------
BaseFloatMatrix::~BaseFloatMatrix()
{
  freeHostData();
  if (mDeviceData && ctx()) kw_free(ctx(), mDeviceData);
  mDeviceData = nullptr;
}
void BaseFloatMatrix::allocate(size_t capacityFloats)
{
  mCapacity = capacityFloats;
  void* d   = nullptr;
  kwCheck(kw_malloc(ctx(), mCapacity * sizeof(float), &d));
  mDeviceData = static_cast<float*>(d);
  kwCheck(kw_memset(ctx(), mDeviceData, 0, mCapacity * sizeof(float))); // reference zero-initialises (:134-146)
}
float* BaseFloatMatrix::getHostData()
{
  if (!mHostData)
  {
    void* p = nullptr;
    if (posix_memalign(&p, 64, mCapacity * sizeof(float)) != 0) throw std::bad_alloc();
    mHostData = static_cast<float*>(p);
    std::memset(mHostData, 0, mCapacity * sizeof(float));
  }
  return mHostData;
}
void BaseFloatMatrix::freeHostData()
{
  std::free(mHostData);
  mHostData = nullptr;
}
void BaseFloatMatrix::copyToDevice()
{
  if (!mHostData) return; // nothing was ever put on the host side: the device copy (zeros) is authoritative
  kwCheck(kw_memcpy_h2d(ctx(), mDeviceData, mHostData, mCapacity * sizeof(float)));
}
void BaseFloatMatrix::copyFromDevice()
{
  kwCheck(kw_memcpy_d2h(ctx(), getHostData(), mDeviceData, mCapacity * sizeof(float)));
}
void BaseFloatMatrix::zeroDeviceMatrix() { kwCheck(kw_memset(ctx(), mDeviceData, 0, mCapacity * sizeof(float))); }
void BaseFloatMatrix::scalarDividedBy(float scalar)
{
  float* d = getHostData();
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < mCapacity; i++) d[i] = scalar / d[i];
}

// ---- RealMatrix / ComplexMatrix -------------------------------------------------------------------------------------
RealMatrix::RealMatrix(const DimensionSizes& dims)
{
  mDimensionSizes = dims;
  mSize           = dims.nx * dims.ny * dims.nz;
  allocate(mSize);
}
void RealMatrix::readData(const InputProvider& in, const std::string& name)
{
  if (in.getDatasetType(name) != InputProvider::DataType::kFloat)
    throw std::ios_base::failure("Error: Matrix [" + name + "] data type is not of single precision floating point");
  in.readFloat(name, getHostData(), mSize);
}
ComplexMatrix::ComplexMatrix(const DimensionSizes& dims)
{
  mDimensionSizes = dims;
  mSize           = dims.nx * dims.ny * dims.nz;
  allocate(2 * mSize);
}
void ComplexMatrix::readData(const InputProvider& in, const std::string& name)
{
  if (in.getDatasetType(name) != InputProvider::DataType::kFloat)
    throw std::ios_base::failure("Error: Matrix [" + name + "] data type is not of single precision floating point");
  // complex datasets are stored with a doubled fastest dimension (ComplexMatrix.cpp:58-88)
  in.readFloat(name, getHostData(), 2 * mSize);
}

// ---- IndexMatrix ----------------------------------------------------------------------------------------------------
IndexMatrix::IndexMatrix(const DimensionSizes& dims) : mDimensionSizes(dims), mSize(dims.nx * dims.ny * dims.nz)
{
  mHostData.assign(mSize, 0);
  void* d = nullptr;
  kwCheck(kw_malloc(ctx(), mSize * sizeof(size_t), &d));
  mDeviceData = static_cast<size_t*>(d);
}
IndexMatrix::~IndexMatrix()
{
  if (mDeviceData && ctx()) kw_free(ctx(), mDeviceData);
}
void IndexMatrix::readData(const InputProvider& in, const std::string& name)
{
  if (in.getDatasetType(name) != InputProvider::DataType::kLong)
    throw std::ios_base::failure("Error: Matrix [" + name + "] data type is not of 64-bit unsigned integer");
  in.readIndex(name, mHostData.data(), mSize);
}
void IndexMatrix::copyToDevice() { kwCheck(kw_memcpy_h2d(ctx(), mDeviceData, mHostData.data(), mSize * sizeof(size_t))); }
void IndexMatrix::copyFromDevice() { kwCheck(kw_memcpy_d2h(ctx(), mHostData.data(), mDeviceData, mSize * sizeof(size_t))); }
void IndexMatrix::recomputeIndicesToCPP()
{
  for (size_t i = 0; i < mSize; i++) mHostData[i]--;
}
void IndexMatrix::recomputeIndicesToMatlab()
{
  for (size_t i = 0; i < mSize; i++) mHostData[i]++;
}
DimensionSizes IndexMatrix::getTopLeftCorner(size_t c) const
{
  return DimensionSizes(mHostData[6 * c], mHostData[6 * c + 1], mHostData[6 * c + 2]);
}
DimensionSizes IndexMatrix::getBottomRightCorner(size_t c) const
{
  return DimensionSizes(mHostData[6 * c + 3], mHostData[6 * c + 4], mHostData[6 * c + 5]);
}
size_t IndexMatrix::getSizeOfCuboid(size_t c) const
{
  const DimensionSizes tl = getTopLeftCorner(c), br = getBottomRightCorner(c);
  return (br.nx - tl.nx + 1) * (br.ny - tl.ny + 1) * (br.nz - tl.nz + 1);
}
size_t IndexMatrix::getSizeOfAllCuboids() const
{
  size_t n = 0;
  for (size_t c = 0; c < mDimensionSizes.ny; c++) n += getSizeOfCuboid(c);
  return n;
}

// ---- HipFftComplexMatrix --------------------------------------------------------------------------------------------
void HipFftComplexMatrix::createR2CFftPlanND(const DimensionSizes&) { kwCheck(kw_fft_create_plans_3d(ctx())); }
void HipFftComplexMatrix::createC2RFftPlanND(const DimensionSizes&) { /* created together with R2C */ }
void HipFftComplexMatrix::createR2CFftPlan1DX(const DimensionSizes&) { kwCheck(kw_fft_create_plans_1d(ctx(), 0)); }
void HipFftComplexMatrix::createR2CFftPlan1DY(const DimensionSizes&) { kwCheck(kw_fft_create_plans_1d(ctx(), 1)); }
void HipFftComplexMatrix::createR2CFftPlan1DZ(const DimensionSizes&) { kwCheck(kw_fft_create_plans_1d(ctx(), 2)); }
void HipFftComplexMatrix::destroyAllPlansAndStaticData() { if (ctx()) kwCheck(kw_fft_destroy_plans(ctx())); }
void HipFftComplexMatrix::computeR2CFftND(RealMatrix& in) { kwCheck(kw_fft_r2c_3d(ctx(), in.getDeviceData(), mDeviceData)); }
void HipFftComplexMatrix::computeC2RFftND(RealMatrix& out) { kwCheck(kw_fft_c2r_3d(ctx(), mDeviceData, out.getDeviceData())); }
void HipFftComplexMatrix::computeR2CFft1DX(RealMatrix& in) { kwCheck(kw_fft_r2c_1d(ctx(), 0, in.getDeviceData(), mDeviceData)); }
void HipFftComplexMatrix::computeR2CFft1DY(RealMatrix& in) { kwCheck(kw_fft_r2c_1d(ctx(), 1, in.getDeviceData(), mDeviceData)); }
void HipFftComplexMatrix::computeR2CFft1DZ(RealMatrix& in) { kwCheck(kw_fft_r2c_1d(ctx(), 2, in.getDeviceData(), mDeviceData)); }
void HipFftComplexMatrix::computeC2RFft1DX(RealMatrix& out) { kwCheck(kw_fft_c2r_1d(ctx(), 0, mDeviceData, out.getDeviceData())); }
void HipFftComplexMatrix::computeC2RFft1DY(RealMatrix& out) { kwCheck(kw_fft_c2r_1d(ctx(), 1, mDeviceData, out.getDeviceData())); }
void HipFftComplexMatrix::computeC2RFft1DZ(RealMatrix& out) { kwCheck(kw_fft_c2r_1d(ctx(), 2, mDeviceData, out.getDeviceData())); }
