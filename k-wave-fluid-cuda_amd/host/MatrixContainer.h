// MatrixContainer.h — enum MatrixIdx -> {matrix, type, dims, load?, checkpoint?, dataset name}.
// Mirror of Containers/MatrixContainer.{h,cpp} + MatrixRecord.{h,cpp} of the reference (enum at
// MatrixContainer.h:63-207, init() at MatrixContainer.cpp:73-411, typed getter at .h:255-269): decides which
// matrices exist for a given medium / source / output selection.
#ifndef KW_HOST_MATRIX_CONTAINER_H
#define KW_HOST_MATRIX_CONTAINER_H
#include <map>
#include <stdexcept>
#include <string>

#include "Matrices.h"

struct MatrixRecord
{
  enum class MatrixType { kReal, kComplex, kIndex, kFft };
  BaseMatrix*    matrixPtr = nullptr;
  MatrixType     matrixType = MatrixType::kReal;
  DimensionSizes dimensionSizes;
  bool           loadData = false;
  bool           checkpoint = false;
  std::string    matrixName;
  void set(MatrixType type, const DimensionSizes& dims, bool load, bool cp, const std::string& name)
  {
    matrixPtr = nullptr; matrixType = type; dimensionSizes = dims; loadData = load; checkpoint = cp; matrixName = name;
  }
};

class MatrixContainer
{
 public:
  enum class MatrixIdx
  {
    kKappa, kSourceKappa, kC2, kP, kRhoX, kRhoY, kRhoZ, kUxSgx, kUySgy, kUzSgz, kDuxdx, kDuydy, kDuzdz, kRho0,
    kDtRho0Sgx, kDtRho0Sgy, kDtRho0Sgz, kDdxKShiftPosR, kDdyKShiftPos, kDdzKShiftPos, kDdxKShiftNegR, kDdyKShiftNeg,
    kDdzKShiftNeg, kPmlXSgx, kPmlYSgy, kPmlZSgz, kPmlX, kPmlY, kPmlZ, kBOnA, kAbsorbTau, kAbsorbEta, kAbsorbNabla1,
    kAbsorbNabla2, kSensorMaskIndex, kSensorMaskCorners, kInitialPressureSourceInput, kPressureSourceInput,
    kTransducerSourceInput, kVelocityXSourceInput, kVelocityYSourceInput, kVelocityZSourceInput, kPressureSourceIndex,
    kVelocitySourceIndex, kDelayMask, kUxShifted, kUyShifted, kUzShifted, kXShiftNegR, kYShiftNegR, kZShiftNegR,
    kTemp1RealND, kTemp2RealND, kTemp3RealND, kTempHipFftX, kTempHipFftY, kTempHipFftZ, kTempHipFftShift,
    kDxudxn, kDyudyn, kDzudzn, kDxudxnSgx, kDyudynSgy, kDzudznSgz
  };

  MatrixContainer() = default;
  ~MatrixContainer() { freeMatrices(); }
  size_t size() const { return mContainer.size(); }
  bool   empty() const { return mContainer.empty(); }
  bool   has(MatrixIdx idx) const { return mContainer.count(idx) != 0; }

  /// typed getter (MatrixContainer.h:255-269)
  template<typename T> T& getMatrix(MatrixIdx idx) const
  {
    auto it = mContainer.find(idx);
    if (it == mContainer.end() || it->second.matrixPtr == nullptr)
      throw std::runtime_error("MatrixContainer: matrix is not allocated for this simulation setup");
    return static_cast<T&>(*(it->second.matrixPtr));
  }
  /// device pointer or nullptr when the matrix does not exist (scalar medium) — what kernel wrappers pass on
  const float* realDeviceOrNull(MatrixIdx idx) const
  {
    auto it = mContainer.find(idx);
    return (it == mContainer.end() || !it->second.matrixPtr) ? nullptr
                                                             : static_cast<BaseFloatMatrix*>(it->second.matrixPtr)->getDeviceData();
  }

  void init();                                   // MatrixContainer.cpp:73-411
  void createMatrices();                         // :418-462
  void freeMatrices();                           // :468-479
  void loadDataFromInputFile(const InputProvider& in); // :485-497
  void copyMatricesToDevice();                   // :544-550
  void copyMatricesFromDevice();
  std::map<MatrixIdx, MatrixRecord>& records() { return mContainer; }

 private:
  std::map<MatrixIdx, MatrixRecord> mContainer;
};
#endif
