// Matrices.h — host+device twin data objects.
// Mirror of MatrixClasses/{BaseMatrix,BaseFloatMatrix,BaseIndexMatrix,RealMatrix,ComplexMatrix,IndexMatrix}.{h,cpp}
// and of MatrixClasses/CufftComplexMatrix.{h,cpp} (here HipFftComplexMatrix, backed by rocFFT through the C-ABI).
// Layout contract: fp32 row-major x-fastest; complex interleaved (re,im) (ComplexMatrix.cpp:124-135);
// indices size_t, converted 1-based -> 0-based by recomputeIndicesToCPP (IndexMatrix.cpp:161-168).
// Host buffers are allocated lazily (the reference allocates both twins eagerly, BaseFloatMatrix.cpp:124-146).
#ifndef KW_HOST_MATRICES_H
#define KW_HOST_MATRICES_H
#include <cstddef>
#include <string>
#include <vector>

#include "DimensionSizes.h"
#include "InputProvider.h"
#include "kwave_hip.h"

class BaseMatrix
{
 public:
  virtual ~BaseMatrix() = default;
  virtual const DimensionSizes& getDimensionSizes() const = 0;
  virtual size_t size() const = 0;
  virtual size_t capacity() const = 0;
  virtual void   readData(const InputProvider& in, const std::string& name) = 0;
  virtual void   copyToDevice() = 0;
  virtual void   copyFromDevice() = 0;
};

class BaseFloatMatrix : public BaseMatrix
{
 public:
  ~BaseFloatMatrix() override;
  const DimensionSizes& getDimensionSizes() const override { return mDimensionSizes; }
  size_t size() const override { return mSize; }
  size_t capacity() const override { return mCapacity; }
  float*       getHostData();
  const float* getHostData() const { return const_cast<BaseFloatMatrix*>(this)->getHostData(); }
  float*       getDeviceData() { return mDeviceData; }
  const float* getDeviceData() const { return mDeviceData; }
  void copyToDevice() override;
  void copyFromDevice() override;
  /// BaseFloatMatrix::zeroDeviceMatrix (BaseFloatMatrix.cpp:77-80), asynchronous on the context's stream
  void zeroDeviceMatrix();
  /// BaseFloatMatrix::scalarDividedBy (BaseFloatMatrix.cpp:86-93): host data = scalar / host data
  void scalarDividedBy(float scalar);
  void freeHostData();

 protected:
  void allocate(size_t capacityFloats);
  DimensionSizes mDimensionSizes;
  size_t mSize = 0, mCapacity = 0;
  float* mHostData   = nullptr;
  float* mDeviceData = nullptr;
};

class RealMatrix : public BaseFloatMatrix
{
 public:
  explicit RealMatrix(const DimensionSizes& dims);
  void readData(const InputProvider& in, const std::string& name) override;
};

class ComplexMatrix : public BaseFloatMatrix
{
 public:
  explicit ComplexMatrix(const DimensionSizes& dims);
  void readData(const InputProvider& in, const std::string& name) override;
};

class IndexMatrix : public BaseMatrix
{
 public:
  explicit IndexMatrix(const DimensionSizes& dims);
  ~IndexMatrix() override;
  const DimensionSizes& getDimensionSizes() const override { return mDimensionSizes; }
  size_t size() const override { return mSize; }
  size_t capacity() const override { return mSize; }
  size_t*       getHostData() { return mHostData.data(); }
  const size_t* getHostData() const { return mHostData.data(); }
  size_t*       getDeviceData() { return mDeviceData; }
  const size_t* getDeviceData() const { return mDeviceData; }
  void readData(const InputProvider& in, const std::string& name) override;
  void copyToDevice() override;
  void copyFromDevice() override;
  void recomputeIndicesToCPP();    // IndexMatrix.cpp:161-168
  void recomputeIndicesToMatlab(); // IndexMatrix.cpp:174-181
  /// cuboid helpers (IndexMatrix.cpp:134-155): corners stored x1,y1,z1,x2,y2,z2 per cuboid
  DimensionSizes getTopLeftCorner(size_t cuboid) const;
  DimensionSizes getBottomRightCorner(size_t cuboid) const;
  size_t         getSizeOfCuboid(size_t cuboid) const;
  size_t         getSizeOfAllCuboids() const;

 private:
  DimensionSizes      mDimensionSizes;
  size_t              mSize = 0;
  std::vector<size_t> mHostData;
  size_t*             mDeviceData = nullptr;
};

/// Replaces class CufftComplexMatrix (MatrixClasses/CufftComplexMatrix.h:50-270): the object *is* the complex buffer,
/// plans are per-context statics shared by all instances.
class HipFftComplexMatrix : public ComplexMatrix
{
 public:
  explicit HipFftComplexMatrix(const DimensionSizes& dims) : ComplexMatrix(dims) {}
  static void createR2CFftPlanND(const DimensionSizes&); // CufftComplexMatrix.cpp:82-100
  static void createC2RFftPlanND(const DimensionSizes&); // :108-130
  static void createR2CFftPlan1DX(const DimensionSizes&);
  static void createR2CFftPlan1DY(const DimensionSizes&);
  static void createR2CFftPlan1DZ(const DimensionSizes&);
  static void createC2RFftPlan1DX(const DimensionSizes&) {}
  static void createC2RFftPlan1DY(const DimensionSizes&) {}
  static void createC2RFftPlan1DZ(const DimensionSizes&) {}
  static void destroyAllPlansAndStaticData(); // :432-502
  void computeR2CFftND(RealMatrix& inMatrix);  // :508-518
  void computeC2RFftND(RealMatrix& outMatrix); // :524-534
  void computeR2CFft1DX(RealMatrix& inMatrix);
  void computeR2CFft1DY(RealMatrix& inMatrix);
  void computeR2CFft1DZ(RealMatrix& inMatrix);
  void computeC2RFft1DX(RealMatrix& outMatrix);
  void computeC2RFft1DY(RealMatrix& outMatrix);
  void computeC2RFft1DZ(RealMatrix& outMatrix);
};
#endif
